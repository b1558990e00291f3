// tests/cxx_threads.cpp -- the reference's threading contract (README.md:52-55: "thread-safe ... a separate object per
// stream"; SURVEY.md 8b "Threading"): objects are not synchronised internally, DIFFERENT objects may be created and
// driven from different host threads at the same time, the designer's caches are shared by all of them.
// Ten threads start together with cold caches: eight create a resampler of a ratio of their own, two more the SAME
// ratio as thread 0 (cache hit / double design under contention), each runs 50 r8b_process calls; every stream must equal
// bit for bit what the same calls give on one thread afterwards.  r8b_last_error() is per thread: a thread that
// provokes an error sees its text, its neighbours see none.  With -DWITH_HIP (GPU tier, hipcc): the same for batch
// objects fed with device pointers on a stream per thread.
// Links against libr8bsrc_hip.so (GPU tier) or the host emulation tests/emul/_build/libr8bsrc_emul.so (CPU tier: the
// host side -- designer, caches, plan, error slots -- is the same code).  Exit code 0 = all equal.
#ifdef WITH_HIP
#include <hip/hip_runtime.h>
#endif

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/r8bsrc.h"

static double splitmix(uint64_t& s)
{
	uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	z ^= z >> 31;
	return (double) (z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

static const double kRates[8][2] = { { 44100, 96000 }, { 96000, 44100 }, { 44100, 88200 }, { 88200, 44100 },
	{ 48000, 32000 }, { 44100, 44101 }, { 44100, 48000 }, { 32000, 48000 } };
static const int kThreads = 10, kCalls = 50, kLen = 1000;

static std::atomic<int> g_ready(0);
static void start_together(int n)
{
	g_ready.fetch_add(1);
	while (g_ready.load() < n) std::this_thread::yield();
}

// one stream through the reference-shaped entries (host buffers); out: every output sample, in order
static int run_single(int ri, std::vector<double>& out, bool threaded)
{
	if (threaded) start_together(kThreads);
	CR8BResampler rs = r8b_create(kRates[ri][0], kRates[ri][1], kLen, 2.0, r8brr24);
	if (rs == nullptr) return 2;
	if (r8b_last_error()[0] != 0) return 8; // (nothing went wrong in THIS thread)
	std::vector<double> in((size_t) kLen);
	uint64_t seed = 100 + (uint64_t) ri;
	for (int c = 0; c < kCalls; c++)
	{
		const int l = c % 7 == 3 ? kLen / 3 : kLen;
		for (int i = 0; i < l; i++) in[(size_t) i] = splitmix(seed);
		double* op = nullptr;
		const int n = r8b_process(rs, in.data(), l, op);
		if (n < 0) return 3;
		out.insert(out.end(), op, op + n);
		if (c == 20 && ri == 1)
		{
			// an error of this thread's own: reported here, invisible to the others
			if (r8b_batch_process_host(nullptr, in.data(), kLen, kLen, in.data(), kLen) >= 0) return 9;
			if (r8b_last_error()[0] == 0) return 10;
		}
		else if (ri != 1 && r8b_last_error()[0] != 0) return 11;
	}
	r8b_delete(rs);
	return 0;
}

#ifdef WITH_HIP
// a batch object per thread, device buffers and a stream of the thread's own
static int run_batch(int ri, std::vector<double>& out, bool threaded)
{
	const int nch = 4;
	if (threaded) start_together(8);
	CR8BBatch b = r8b_batch_create(kRates[ri][0], kRates[ri][1], kLen, 2.0, 180.15, nch, -1);
	if (b == nullptr) return 2;
	const int cap = r8b_batch_max_out_len(b);
	hipStream_t stream;
	double *d_in, *d_out;
	if (hipStreamCreate(&stream) != hipSuccess || hipMalloc(&d_in, sizeof(double) * kLen * nch) != hipSuccess ||
		hipMalloc(&d_out, sizeof(double) * (size_t) cap * nch) != hipSuccess) return 4;
	std::vector<double> in((size_t) kLen * nch), o((size_t) cap * nch);
	uint64_t seed = 200 + (uint64_t) ri;
	for (int c = 0; c < kCalls; c++)
	{
		for (size_t i = 0; i < in.size(); i++) in[i] = splitmix(seed);
		if (hipMemcpyAsync(d_in, in.data(), sizeof(double) * in.size(), hipMemcpyHostToDevice, stream) != hipSuccess) return 5;
		const int n = r8b_batch_process(b, d_in, kLen, kLen, d_out, cap, stream);
		if (n < 0) return 3;
		if (hipMemcpyAsync(o.data(), d_out, sizeof(double) * o.size(), hipMemcpyDeviceToHost, stream) != hipSuccess) return 6;
		if (hipStreamSynchronize(stream) != hipSuccess) return 7;
		for (int ch = 0; ch < nch; ch++) out.insert(out.end(), &o[(size_t) ch * cap], &o[(size_t) ch * cap] + n);
	}
	r8b_batch_delete(b);
	(void) hipFree(d_in);
	(void) hipFree(d_out);
	(void) hipStreamDestroy(stream);
	return 0;
}
#endif

template<class F>
static int run_all(const char* what, int nthreads, F f)
{
	std::vector<std::vector<double>> got((size_t) nthreads), want((size_t) nthreads);
	std::vector<int> rc((size_t) nthreads, -1);
	g_ready.store(0);
	std::vector<std::thread> th;
	for (int t = 0; t < nthreads; t++)
		th.emplace_back([&, t]() { rc[(size_t) t] = f(t % 8, got[(size_t) t], true); });
	for (std::thread& t : th) t.join();
	for (int t = 0; t < nthreads; t++)
	{
		if (rc[(size_t) t] != 0)
		{
			fprintf(stderr, "%s: thread %d failed with %d (%s)\n", what, t, rc[(size_t) t], r8b_last_error());
			return 20 + rc[(size_t) t];
		}
		const int r1 = f(t % 8, want[(size_t) t], false);
		if (r1 != 0) return 40 + r1;
		if (got[(size_t) t].size() != want[(size_t) t].size() || got[(size_t) t].empty() ||
			memcmp(got[(size_t) t].data(), want[(size_t) t].data(), sizeof(double) * want[(size_t) t].size()) != 0)
		{
			fprintf(stderr, "%s: thread %d (%g -> %g) differs from the single-threaded run (%zu / %zu samples)\n", what, t,
				kRates[t % 8][0], kRates[t % 8][1], got[(size_t) t].size(), want[(size_t) t].size());
			return 60;
		}
		printf("%s: thread %d (%g -> %g): %zu samples equal\n", what, t, kRates[t % 8][0], kRates[t % 8][1],
			want[(size_t) t].size());
	}
	return 0;
}

int main()
{
	int rc = run_all("r8b_process", kThreads, run_single);
	if (rc != 0) return rc;
#ifdef WITH_HIP
	rc = run_all("r8b_batch_process", 8, run_batch);
	if (rc != 0) return rc;
#endif
	printf("OK\n");
	return 0;
}
