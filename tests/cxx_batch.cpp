// tests/cxx_batch.cpp -- compiled (hipcc) and run by tests/test_gpu_parity.py::test_cxx_batch_device:
// the batch C ABI used the way a C/C++ host application would, with its own hipMalloc'ed buffers and
// its own stream -- no Python, no torch.  Every channel gets the same input, so every output row must
// equal what the reference-shaped single-stream entry (r8b_process, host buffers) returns; the
// checkpoint entries are exercised on the way.  Exit code 0 = all equal.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/r8bsrc.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 10; } } while (0)

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
	const int nch = 5, L = 2000, calls = 7;
	const long long in_stride = L + 24; // rows need not be dense
	CR8BBatch b = r8b_batch_create(44100.0, 96000.0, L, 2.0, 180.15, nch, -1);
	CR8BBatch b2 = r8b_batch_create(44100.0, 96000.0, L, 2.0, 180.15, nch, -1);
	CR8BResampler one = r8b_create(44100.0, 96000.0, L, 2.0, r8brr24);
	if (b == nullptr || b2 == nullptr || one == nullptr)
	{
		fprintf(stderr, "create failed: %s\n", r8b_last_error());
		return 2;
	}
	const int cap = r8b_batch_max_out_len(b);
	const long long out_stride = cap + 8;
	hipStream_t stream;
	CHECK(hipStreamCreate(&stream));
	double *d_in, *d_out;
	CHECK(hipMalloc(&d_in, sizeof(double) * in_stride * nch));
	CHECK(hipMalloc(&d_out, sizeof(double) * out_stride * nch));
	std::vector<double> in((size_t) L), rows((size_t) in_stride * nch), out((size_t) out_stride * nch);
	std::vector<unsigned char> blob;
	uint64_t seed = 3;
	for (int c = 0; c < calls; c++)
	{
		for (int i = 0; i < L; i++) in[(size_t) i] = splitmix(seed);
		for (int ch = 0; ch < nch; ch++)
			memcpy(&rows[(size_t) (ch * in_stride)], in.data(), sizeof(double) * L);
		CHECK(hipMemcpyAsync(d_in, rows.data(), sizeof(double) * in_stride * nch,
			hipMemcpyHostToDevice, stream));
		CR8BBatch cur = c < 4 ? b : b2; // calls 4.. continue in b2 from b's checkpoint
		if (c == 4)
		{
			blob.resize((size_t) r8b_batch_state_size(b));
			if (r8b_batch_state_save(b, blob.data(), (long long) blob.size(), stream) < 0) return 6;
			if (r8b_batch_state_load(b2, blob.data(), (long long) blob.size(), stream) != 0) return 7;
		}
		const int n = r8b_batch_process(cur, d_in, in_stride, L, d_out, out_stride, stream);
		if (n < 0)
		{
			fprintf(stderr, "process failed: %s\n", r8b_last_error());
			return 3;
		}
		CHECK(hipMemcpyAsync(out.data(), d_out, sizeof(double) * out_stride * nch,
			hipMemcpyDeviceToHost, stream));
		CHECK(hipStreamSynchronize(stream));
		double* ref;
		const int n1 = r8b_process(one, in.data(), L, ref);
		if (n1 != n) return 4;
		// The pair kernel carries channels 2c and 2c+1 in the real and imaginary part of one complex
		// transform; the single-stream object rides in the real part with a copy of itself beside it.
		// So even rows (and the unpaired last one) equal the single-stream result bit for bit, odd rows
		// to rounding (a few 1e-16 of full scale).
		for (int ch = 0; ch < nch; ch++)
			for (int i = 0; i < n; i++)
			{
				const double d = out[(size_t) (ch * out_stride + i)] - ref[i];
				if ((ch & 1) == 0 ? d != 0.0 : (d > 4e-15 || d < -4e-15))
				{
					fprintf(stderr, "call %d, channel %d, output %d of %d: batch %.17g, single stream %.17g (difference %.3g)\n",
						c, ch, i, n, out[(size_t) (ch * out_stride + i)], ref[i], d);
					return 5;
				}
			}
		printf("call %d: %d samples x %d channels equal\n", c, n, nch);
	}
	r8b_delete(one);
	r8b_batch_delete(b);
	r8b_batch_delete(b2);
	CHECK(hipFree(d_in));
	CHECK(hipFree(d_out));
	CHECK(hipStreamDestroy(stream));
	printf("OK\n");
	return 0;
}
