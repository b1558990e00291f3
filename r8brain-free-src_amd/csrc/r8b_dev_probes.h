// r8b_dev_probes.h -- DEVELOPMENT BUILDS ONLY (-DR8B_CP_STAMPS=<first workgroup> / -DR8B_TIMELINE, tools/variant.sh):
// device buffers and host accessors of the pair kernel's probes (tools/stamps_probe.py, tools/timeline_probe.py).
// Included by r8b_kernels.hip inside namespace r8bhip { namespace { ... } } behind those macros; the shipped library
// does not compile it.
#ifdef R8B_CP_STAMPS
__device__ long long g_cp_stamps[8 * 8 * 32];
} // namespace
} // namespace r8bhip
// (development builds: the stamps of workgroups R8B_CP_STAMPS .. + 7 of the last launch; tools/stamps_probe.py)
extern "C" __attribute__((visibility("default"))) void r8b_dev_stamps(long long* out)
{
	(void) hipDeviceSynchronize();
	(void) hipMemcpyFromSymbol(out, HIP_SYMBOL(r8bhip::g_cp_stamps), sizeof(long long) * 8 * 8 * 32);
}
namespace r8bhip {
namespace {
#endif
#ifdef R8B_TIMELINE
// (development, tools/timeline_probe.py: every workgroup of the last pair-kernel launch leaves where and when it ran --
// hardware id, cycle counter at its first instruction, at the arrival of its samples and at its last store's issue)
__device__ long long g_timeline[16384 * 4];
} // namespace
} // namespace r8bhip
extern "C" __attribute__((visibility("default"))) void r8b_dev_timeline(long long* out, int n)
{
	(void) hipDeviceSynchronize();
	(void) hipMemcpyFromSymbol(out, HIP_SYMBOL(r8bhip::g_timeline), sizeof(long long) * 4 * (size_t) n);
}
namespace r8bhip {
namespace {
#endif
