// tests/cxx_frontend.cpp -- compiled and run by tests/test_gpu_parity.py::test_cxx_frontend: the
// reference's usage pattern (example.cpp:47-67) against include/r8b/CDSPResampler.h and, through
// plain C calls, the five DLL symbols.  Prints the output stream as hex doubles.
#include <cstdio>
#include <cstdint>
#include <vector>

#include "../include/r8b/CDSPResampler.h"

static double splitmix(uint64_t& s)
{
	uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	z ^= z >> 31;
	return (double) (z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

int main()
{
	const int L = 1024, calls = 6;
	uint64_t seed = 7;
	std::vector<double> in((size_t) L);
	r8b::CDSPResampler24 rs(44100.0, 96000.0, L);
	CR8BResampler dll = r8b_create(44100.0, 96000.0, L, 2.0, r8brr24);
	if (dll == nullptr) return 2;
	printf("inlen %d %d\n", rs.getInputRequiredForOutput(1), r8b_inlen(dll, 1));
	// getLatencyFrac (reference CDSPResampler.h:491-494): 0 for linear phase, the last stage's residual for fprMinPhase
	{
		const double rates[8][2] = { { 44100.0, 88200.0 }, { 44100.0, 96000.0 }, { 96000.0, 44100.0 }, { 44100.0, 44101.0 },
			{ 44100.0, 176400.0 }, { 176400.0, 44100.0 }, { 88200.0, 44100.0 }, { 48000.0, 32000.0 } };
		for (int i = 0; i < 8; i++)
		{
			r8b::CDSPResampler lin(rates[i][0], rates[i][1], L, 2.0, 180.15, r8b::fprLinearPhase);
			r8b::CDSPResampler mp(rates[i][0], rates[i][1], L, 2.0, 180.15, r8b::fprMinPhase);
			printf("latfrac %.1f %.1f %a %a\n", rates[i][0], rates[i][1], lin.getLatencyFrac(), mp.getLatencyFrac());
		}
	}
	for (int c = 0; c < calls; c++)
	{
		for (int i = 0; i < L; i++) in[(size_t) i] = splitmix(seed);
		double* op;
		const int n = rs.process(in.data(), L, op);
		double* op2;
		const int n2 = r8b_process(dll, in.data(), L, op2);
		printf("call %d %d %d\n", c, n, n2);
		for (int i = 0; i < n; i++)
		{
			if (op[i] != op2[i]) return 3; // both entries run the same kernels: bitwise equal
			printf("%a\n", op[i]);
		}
	}
	r8b_delete(dll);
	return 0;
}
