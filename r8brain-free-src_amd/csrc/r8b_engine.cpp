// r8b_engine.cpp -- see r8b_engine.h.  Host C++ only; every device operation goes through
// r8b_launch.h.
#include "r8b_engine.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <list>
#include <mutex>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace r8bhip {

std::vector<double> make_twiddles(int len)
{
	std::vector<double> t((size_t) len * 2);
	const long double two_pi = 6.283185307179586476925286766559L;
	const int quarter = len / 4;
	for (int e = 0; e < len; e++)
	{
		double c, s;
		if (quarter > 0 && len % 4 == 0)
		{
			const int quad = e / quarter, rem = e - quad * quarter;
			long double c0 = 1.0L, s0 = 0.0L;
			if (rem != 0)
			{
				// evaluate in the first octant pair and reflect for accuracy/symmetry
				if (2 * rem <= quarter)
				{
					c0 = cosl(two_pi * rem / len);
					s0 = sinl(two_pi * rem / len);
				}
				else
				{
					const int r2 = quarter - rem;
					c0 = sinl(two_pi * r2 / len);
					s0 = cosl(two_pi * r2 / len);
				}
			}
			switch (quad)
			{
			case 0: c = (double) c0; s = (double) s0; break;
			case 1: c = (double) -s0; s = (double) c0; break;
			case 2: c = (double) -c0; s = (double) -s0; break;
			default: c = (double) s0; s = (double) -c0; break;
			}
		}
		else
		{
			c = (double) cosl(two_pi * e / len);
			s = (double) sinl(two_pi * e / len);
		}
		t[(size_t) e * 2] = c;
		t[(size_t) e * 2 + 1] = -s;
	}
	return t;
}

std::vector<double> kernel_spectrum(const LpFilter& f, int bl2, double scale)
{
	const std::vector<double> tw = make_twiddles(bl2);
	std::vector<double> H((size_t) bl2 / 2 + 1);
	const double* c = &f.taps[(size_t) f.fl2];
	for (int m = 0; m <= bl2 / 2; m++)
	{
		long double acc = c[0];
		for (int t = 1; t <= f.fl2; t++)
		{
			const int e = (int) (((long long) m * t) & (bl2 - 1));
			acc += 2.0L * c[t] * tw[(size_t) e * 2];
		}
		H[(size_t) m] = (double) (acc * scale);
	}
	return H;
}

std::vector<double> kernel_spectrum_complex(const LpFilter& f, int bl2, int align, double scale)
{
	const std::vector<double> tw = make_twiddles(bl2);
	std::vector<double> H((size_t) (bl2 / 2 + 1) * 2);
	const int K = (int) f.taps.size();
	for (int m = 0; m <= bl2 / 2; m++)
	{
		long double re = 0.0L, im = 0.0L;
		for (int n = 0; n < K; n++)
		{
			const long long e = ((long long) m * (n - align)) & (bl2 - 1);
			re += (long double) f.taps[(size_t) n] * tw[(size_t) e * 2];
			im += (long double) f.taps[(size_t) n] * tw[(size_t) e * 2 + 1];
		}
		H[(size_t) m * 2] = (double) (re * scale);
		H[(size_t) m * 2 + 1] = (double) (im * scale);
	}
	return H;
}

std::vector<double> spectral_constants(const std::vector<double>& H, const std::vector<double>& tw,
	int bl2, int n_in, int up)
{
	typedef std::complex<long double> C;
	const int N = n_in / 2, N2 = N * up;
	const int slots = N / 2 + 1;
	int logn = 0;
	while ((1 << logn) < N) logn++;
	const int nconst = up == 1 ? 4 : 8;
	std::vector<double> out((size_t) nconst * slots * 2, 0.0);
	auto W = [&](long long e) // exp(-2 pi i e / bl2)
	{
		e &= bl2 - 1;
		return C(tw[(size_t) e * 2], tw[(size_t) e * 2 + 1]);
	};
	auto put = [&](int c, int slot, C v)
	{
		out[((size_t) c * slots + slot) * 2] = (double) v.real();
		out[((size_t) c * slots + slot) * 2 + 1] = (double) v.imag();
	};
	const C I(0.0L, 1.0L);
	const int tsf = bl2 / (2 * N), tsh = bl2 / (2 * N2);
	for (int slot = 0; slot < slots; slot++)
	{
		int kf = N / 2;
		if (slot < N / 2)
		{
			kf = 0;
			for (int b = 0; b < logn - 1; b++)
				if (slot & (1 << b)) kf |= 1 << (logn - 2 - b);
		}
		// R[kf] = A Z1 + B conj(Z2),  R[N-kf] = conj(B) conj(Z1) + conj(A) Z2
		const C w = W((long long) kf * tsf);
		const C A = 0.5L * (C(1.0L) - I * w), B = 0.5L * (C(1.0L) + I * w);
		auto cw = [&](int k) { return std::conj(W((long long) k * tsh)); }; // conj(w_{2 N2}^k)
		if (up == 1)
		{
			const long double ha = H[(size_t) kf], hb = H[(size_t) (N - kf)];
			// Z'[kf] = ha (1 + i cw) R + hb (1 - i cw) conj(R[N-kf])
			C f = ha * (C(1.0L) + I * cw(kf)), g = hb * (C(1.0L) - I * cw(kf));
			put(0, slot, f * A + g * B);
			put(1, slot, f * B + g * A);
			// Z'[N-kf] = hb (1 + i cw2) R[N-kf] + ha (1 - i cw2) conj(R)
			f = hb * (C(1.0L) + I * cw(N - kf));
			g = ha * (C(1.0L) - I * cw(N - kf));
			put(2, slot, f * std::conj(B) + g * std::conj(A));
			put(3, slot, f * std::conj(A) + g * std::conj(B));
		}
		else
		{
			// zero-stuffed spectrum: bins kf / 2N-kf carry R / conj R, bins N-kf / N+kf carry
			// R[N-kf] / its conjugate
			long double ha = H[(size_t) kf], hb = H[(size_t) (2 * N - kf)];
			C c1 = C(ha + hb) + I * cw(kf) * C(ha - hb);
			put(0, slot, c1 * A);
			put(1, slot, c1 * B);
			C d1 = C(hb + ha) + I * cw(2 * N - kf) * C(hb - ha);
			put(2, slot, d1 * std::conj(A));
			put(3, slot, d1 * std::conj(B));
			if (kf != 0)
			{
				ha = H[(size_t) (N - kf)];
				hb = H[(size_t) (N + kf)];
				C c2 = C(ha + hb) + I * cw(N - kf) * C(ha - hb);
				put(4, slot, c2 * std::conj(B));
				put(5, slot, c2 * std::conj(A));
				C d2 = C(hb + ha) + I * cw(N + kf) * C(hb - ha);
				put(6, slot, d2 * B);
				put(7, slot, d2 * A);
			}
			else
			{
				// bin N is its own partner: Z'[N] = 2 H[N] R[N]
				const C c2 = C(2.0L * H[(size_t) N]);
				put(4, slot, c2 * std::conj(B));
				put(5, slot, c2 * std::conj(A));
			}
		}
	}
	return out;
}

// The same for the 2x-decimating convolver (r8b_convx.h, UPLOG < 0): slot -> backward pair
// (k, N2-k), N2 = N / down, fed by the forward bins idx = {k, N-k, N2-k, N-N2+k}:
//   Z'[k]    = c0 Z1 + c1 conj Z2 + c2 conj Z3 + c3 Z4      (form 1)
//   Z'[N2-k] = c4 conj Z1 + c5 Z2 + c6 Z3 + c7 conj Z4      (form 2; slot k = 0: added to Z'[0])
// The constants are read off a long-double model of the chain real-FFT unpacking -> kernel
// multiplication -> Nyquist fix-up of the shortened transform (reference
// CDSPBlockConvolver.h:329-342) -> packing, by probing it with unit inputs: a real-linear map of
// one complex input is a*z + b*conj(z) with a = (F(1) - i F(i)) / 2, b = (F(1) + i F(i)) / 2.
std::vector<double> spectral_constants_down(const std::vector<double>& H,
	const std::vector<double>& tw, int bl2, int n_in, int down)
{
	typedef std::complex<long double> C;
	const int N = n_in / 2, N2 = N / down;
	const int slots = N2 / 2 + 1;
	int logn2 = 0;
	while ((1 << logn2) < N2) logn2++;
	std::vector<double> out((size_t) 8 * slots * 2, 0.0);
	const C I(0.0L, 1.0L);
	auto W = [&](long long e) // exp(-2 pi i e / bl2)
	{
		e &= bl2 - 1;
		return C(tw[(size_t) e * 2], tw[(size_t) e * 2 + 1]);
	};
	long double hmax = 0.0L;
	for (double h : H) hmax = std::max(hmax, (long double) std::fabs(h));
	for (int slot = 0; slot < slots; slot++)
	{
		int k = N2 / 2;
		if (slot < N2 / 2)
		{
			k = 0;
			for (int b = 0; b < logn2 - 1; b++)
				if (slot & (1 << b)) k |= 1 << (logn2 - 2 - b);
		}
		const int idx[4] = { k, (N - k) & (N - 1), N2 - k, (N - N2 + k) & (N - 1) };
		// the model, with the forward transform being zero except Z[pe] = pv
		int pe = 0;
		C pv;
		auto Z = [&](int j) { return j == pe ? pv : C(0.0L); };
		auto X = [&](int m) // bin m of the bl2-point real spectrum, 0 <= m <= N
		{
			const C z1 = Z(m & (N - 1)), z2 = std::conj(Z((N - m) & (N - 1)));
			return 0.5L * (z1 + z2) + W((long long) m * (bl2 / (2 * N))) * ((z1 - z2) / (2.0L * I));
		};
		auto Y = [&](int m) // product spectrum handed to the backward transform, 0 <= m <= N2
		{
			const C x = X(m);
			if (m == N2) return C((long double) H[(size_t) m] * (x.real() + x.imag()));
			return (long double) H[(size_t) m] * x;
		};
		auto Zp = [&](int m) // packed input m of the N2-point backward transform
		{
			const C sa = Y(m), sb = std::conj(Y(N2 - m));
			return (sa + sb) + I * std::conj(W((long long) m * (bl2 / (2 * N2)))) * (sa - sb);
		};
		const int nout = k != 0 && k != N2 / 2 ? 2 : 1;
		for (int o = 0; o < nout; o++)
		{
			const int m = o == 0 ? k : N2 - k;
			for (int u = 0; u < 4; u++)
			{
				bool seen = false;
				for (int v = 0; v < u; v++) seen = seen || idx[v] == idx[u];
				if (seen) continue;
				pe = idx[u];
				pv = C(1.0L);
				const C f1 = Zp(m);
				pv = I;
				const C fi = Zp(m);
				const C coef[2] = { 0.5L * (f1 - I * fi), 0.5L * (f1 + I * fi) }; // plain, conj
				for (int cj = 0; cj < 2; cj++)
				{
					if (std::abs(coef[cj]) <= 1e-19L * hmax) continue;
					// where does element idx[u] appear plain (cj = 0) / conjugated (cj = 1)?
					// form 1: plain at positions 0, 3; form 2: plain at positions 1, 2
					int where = -1;
					for (int form = 0; form < 2 && where < 0; form++)
					{
						if (k != 0 && form != o) continue;
						for (int pos = 0; pos < 4 && where < 0; pos++)
						{
							const bool plain = form == 0 ? (pos == 0 || pos == 3) :
								(pos == 1 || pos == 2);
							if (idx[pos] == idx[u] && plain == (cj == 0)) where = form * 4 + pos;
						}
					}
					if (where < 0)
						throw std::logic_error("spectral_constants_down: term does not fit");
					out[((size_t) where * slots + slot) * 2] = (double) coef[cj].real();
					out[((size_t) where * slots + slot) * 2 + 1] = (double) coef[cj].imag();
				}
			}
		}
	}
	return out;
}

// Backward bin k = ca(k) Z[k mod N] + cb(k) conj(Z[-k mod N]): every case of the per-slot table `sc`
// (spectral_constants, up 1 or 2) reduces to this form; (ca, cb) of bin k out of that table.
static void bin_constants(const std::vector<double>& sc, int N, int up, int k, double* ca, double* cb)
{
	const int slots = N / 2 + 1;
	int logn = 0;
	while ((1 << logn) < N) logn++;
	auto slot_of = [&](int kf)
	{
		if (kf >= N / 2) return N / 2;
		int s = 0;
		for (int b = 0; b < logn - 1; b++)
			if (kf & (1 << b)) s |= 1 << (logn - 2 - b);
		return s;
	};
	int a, b, kf;
	if (up == 1)
	{
		if (k <= N / 2) { kf = k; a = 0; b = 1; }
		else { kf = N - k; a = 3; b = 2; }
	}
	else
	{
		if (k <= N / 2) { kf = k; a = 0; b = 1; }
		else if (k <= N) { kf = N - k; a = 5; b = 4; }
		else if (k < N + N / 2) { kf = k - N; a = 6; b = 7; }
		else { kf = 2 * N - k; a = 3; b = 2; }
	}
	const size_t ia = ((size_t) a * slots + slot_of(kf)) * 2, ib = ((size_t) b * slots + slot_of(kf)) * 2;
	ca[0] = sc[ia]; ca[1] = sc[ia + 1];
	cb[0] = sc[ib]; cb[1] = sc[ib + 1];
}

// Constants of the fast path's spectral stage addressed by output position (r8b_convx.h
// cx_spec2_compute): entry c * N2 + P holds ca (c = 0) / cb (c = 1) of bin bitrev(P).
std::vector<double> spectral_constants_by_position(const std::vector<double>& sc, int n_in, int up)
{
	const int N = n_in / 2, N2 = N * up;
	int logn2 = 0;
	while ((1 << logn2) < N2) logn2++;
	std::vector<double> out((size_t) N2 * 2 * 2, 0.0);
	for (int P = 0; P < N2; P++)
	{
		int k = 0;
		for (int b = 0; b < logn2; b++)
			if (P & (1 << b)) k |= 1 << (logn2 - 1 - b);
		bin_constants(sc, N, up, k, &out[(size_t) P * 2], &out[((size_t) N2 + P) * 2]);
	}
	return out;
}

std::vector<double> pair_constants(const std::vector<double>& H, int n_in, int n_out)
{
	const int N = n_in, N2 = n_out, NT = std::max(N, N2) / 16;
	int ln = 0;
	while ((1 << ln) < N) ln++;
	const int NH = std::max(N, N2);
	auto Hf = [&](int m) -> long double { return H[(size_t) (m <= NH / 2 ? m : NH - m)]; };
	auto rev = [](int v, int bits)
	{
		int r = 0;
		for (int b = 0; b < bits; b++)
			if (v & (1 << b)) r |= 1 << (bits - 1 - b);
		return r;
	};
	std::vector<double> out((size_t) 8 * NT * 2, 0.0);
	if (N2 < N)
	{
		// decimating form (r8b_convp.h cp_middle_compute_down): thread t keeps the forward positions
		// 16 t + 2 D (c' / 2) + (c' odd ? 2 D - 1 : 0), c' = 0 .. 16 / D - 1; entry (c, t) = H of c' = 2c, 2c + 1
		const int D = N / N2, E2 = 16 / D;
		for (int t = 0; t < NT; t++)
			for (int cp = 0; cp < E2; cp++)
			{
				const int p = 16 * t + 2 * D * (cp >> 1) + ((cp & 1) ? 2 * D - 1 : 0);
				out[((size_t) (cp >> 1) * NT + t) * 2 + (cp & 1)] = (double) Hf(rev(p, ln));
			}
		return out;
	}
	const int up = N2 / N;
	for (int t = 0; t < NT; t++)
		for (int c = 0; c < 8; c++)
		{
			double* o = &out[((size_t) c * NT + t) * 2];
			if (up == 2)
			{
				const int k = rev(8 * t + c, ln);
				o[0] = (double) (Hf(k) + Hf(k + N));
				o[1] = (double) (Hf(k) - Hf(k + N));
			}
			else
			{
				o[0] = (double) Hf(rev(16 * t + 2 * c, ln));
				o[1] = (double) Hf(rev(16 * t + 2 * c + 1, ln));
			}
		}
	return out;
}

std::vector<double> pair_constants_split(const std::vector<double>& H, int n_in)
{
	const int N = n_in, NT = N / 16, NH = 2 * N;
	int ln = 0;
	while ((1 << ln) < N) ln++;
	auto Hf = [&](int m) -> long double { return H[(size_t) (m <= NH / 2 ? m : NH - m)]; };
	auto rev = [](int v, int bits)
	{
		int r = 0;
		for (int b = 0; b < bits; b++)
			if (v & (1 << b)) r |= 1 << (bits - 1 - b);
		return r;
	};
	std::vector<double> out((size_t) 16 * NT * 2, 0.0);
	for (int t = 0; t < NT; t++)
		for (int c = 0; c < 16; c++)
		{
			const int k = rev(16 * t + c, ln);
			out[((size_t) c * NT + t) * 2] = (double) (Hf(k) + Hf(k + N));
			out[((size_t) c * NT + t) * 2 + 1] = (double) (Hf(k) - Hf(k + N));
		}
	return out;
}

std::vector<double> pair_constants_solo(const std::vector<double>& H, int n)
{
	const int N = n, NT = N / 16, NH = 2 * N;
	int ln = 0;
	while ((1 << ln) < N) ln++;
	auto Hf = [&](int m) -> long double { return H[(size_t) (m <= NH / 2 ? m : NH - m)]; };
	auto rev = [](int v, int bits)
	{
		int r = 0;
		for (int b = 0; b < bits; b++)
			if (v & (1 << b)) r |= 1 << (bits - 1 - b);
		return r;
	};
	const long double pi = 3.14159265358979323846264338327950288L;
	std::vector<double> out((size_t) 16 * NT * 2, 0.0);
	for (int t = 0; t < NT; t++)
		for (int c = 0; c < 16; c++)
		{
			const int k = rev(16 * t + c, ln);
			const long double hs = Hf(k) + Hf(k + N), hd = Hf(k) - Hf(k + N), th = pi * k / N;
			out[((size_t) c * NT + t) * 2] = (double) (hs - hd * sinl(th));
			out[((size_t) c * NT + t) * 2 + 1] = (double) (hd * cosl(th));
		}
	return out;
}

std::vector<double> pair_constants_solo_down(const std::vector<double>& H, int n, int down)
{
	const int N = n, NT = N / 16, N2 = N / down;
	int ln = 0;
	while ((1 << ln) < N) ln++;
	auto rev = [](int v, int bits)
	{
		int r = 0;
		for (int b = 0; b < bits; b++)
			if (v & (1 << b)) r |= 1 << (bits - 1 - b);
		return r;
	};
	const long double pi = 3.14159265358979323846264338327950288L;
	std::vector<double> out((size_t) 16 * NT * 2, 0.0);
	for (int t = 0; t < NT; t++)
		for (int c = 0; c < 16; c += down)
		{
			const int k = rev(16 * t + c, ln); // (< N2: the lowest bits of c are the highest of k)
			const long double th = pi * k / N;
			out[((size_t) c * NT + t) * 2] = H[(size_t) k];
			out[((size_t) c * NT + t) * 2 + 1] = H[(size_t) (N2 - k)];
			out[((size_t) (c + 1) * NT + t) * 2] = (double) cosl(th);
			out[((size_t) (c + 1) * NT + t) * 2 + 1] = (double) sinl(th);
		}
	return out;
}

static std::vector<double> long_block_constants_complex(const std::vector<double>& Hc, int n, bool solo)
{
	typedef std::complex<long double> C;
	const int N = n, NT = N / 16, NH = 2 * N;
	int ln = 0;
	while ((1 << ln) < N) ln++;
	auto Hf = [&](int m) -> C
	{
		m &= NH - 1;
		if (m <= NH / 2) return C(Hc[(size_t) m * 2], Hc[(size_t) m * 2 + 1]);
		return std::conj(C(Hc[(size_t) (NH - m) * 2], Hc[(size_t) (NH - m) * 2 + 1]));
	};
	auto rev = [](int v, int bits)
	{
		int r = 0;
		for (int b = 0; b < bits; b++)
			if (v & (1 << b)) r |= 1 << (bits - 1 - b);
		return r;
	};
	const long double pi = 3.14159265358979323846264338327950288L;
	std::vector<double> out((size_t) 32 * NT * 2, 0.0);
	auto put = [&](int row, int t, C v)
	{
		out[((size_t) row * NT + t) * 2] = (double) v.real();
		out[((size_t) row * NT + t) * 2 + 1] = (double) v.imag();
	};
	for (int t = 0; t < NT; t++)
		for (int c = 0; c < 16; c++)
		{
			const int k = rev(16 * t + c, ln);
			const C hs = Hf(k) + Hf(k + N), hd = Hf(k) - Hf(k + N);
			const long double th = pi * k / N;
			if (solo)
			{
				put(c, t, hs - hd * sinl(th));
				put(16 + c, t, C(0.0L, 1.0L) * hd * cosl(th));
			}
			else
			{
				put(c, t, hs);
				put(16 + c, t, hd * C(cosl(th), sinl(th)));
			}
		}
	return out;
}
std::vector<double> pair_constants_solo_down_complex(const std::vector<double>& Hc, int n)
{
	const int N = n, NT = N / 16, N2 = N / 2;
	int ln = 0;
	while ((1 << ln) < N) ln++;
	auto rev = [](int v, int bits)
	{
		int r = 0;
		for (int b = 0; b < bits; b++)
			if (v & (1 << b)) r |= 1 << (bits - 1 - b);
		return r;
	};
	const long double pi = 3.14159265358979323846264338327950288L;
	std::vector<double> out((size_t) 24 * NT * 2, 0.0);
	for (int t = 0; t < NT; t++)
		for (int c = 0; c < 16; c += 2)
		{
			const int k = rev(16 * t + c, ln), j = N2 - k; // (0 <= k < N2: both bins inside the stored half)
			const long double th = pi * k / N;
			out[((size_t) c * NT + t) * 2] = Hc[(size_t) k * 2];
			out[((size_t) c * NT + t) * 2 + 1] = Hc[(size_t) k * 2 + 1];
			out[((size_t) (c + 1) * NT + t) * 2] = (double) cosl(th);
			out[((size_t) (c + 1) * NT + t) * 2 + 1] = (double) sinl(th);
			out[((size_t) (16 + c / 2) * NT + t) * 2] = Hc[(size_t) j * 2];
			out[((size_t) (16 + c / 2) * NT + t) * 2 + 1] = Hc[(size_t) j * 2 + 1];
		}
	return out;
}
std::vector<double> pair_constants_split_complex(const std::vector<double>& Hc, int n)
{
	return long_block_constants_complex(Hc, n, false);
}
std::vector<double> pair_constants_solo_complex(const std::vector<double>& Hc, int n)
{
	return long_block_constants_complex(Hc, n, true);
}

// The same with a complex kernel spectrum Hc (bl2/2 + 1 complex bins, Hermitian beyond): one complex value
// per entry (r8b_convp.h cp_hp_prefetch, CX) -- 1:1: H of backward position 16 t + c; 2x up: Hs (c < 8) / Hd
// (c >= 8) of forward position 8 t + (c & 7); decimating: H of the thread's kept position c.
std::vector<double> pair_constants_complex(const std::vector<double>& Hc, int n_in, int n_out)
{
	typedef std::complex<long double> C;
	const int N = n_in, N2 = n_out, NT = std::max(N, N2) / 16, NH = std::max(N, N2);
	int ln = 0;
	while ((1 << ln) < N) ln++;
	auto Hf = [&](int m) -> C
	{
		m &= NH - 1;
		if (m <= NH / 2) return C(Hc[(size_t) m * 2], Hc[(size_t) m * 2 + 1]);
		return std::conj(C(Hc[(size_t) (NH - m) * 2], Hc[(size_t) (NH - m) * 2 + 1]));
	};
	auto rev = [](int v, int bits)
	{
		int r = 0;
		for (int b = 0; b < bits; b++)
			if (v & (1 << b)) r |= 1 << (bits - 1 - b);
		return r;
	};
	std::vector<double> out((size_t) 16 * NT * 2, 0.0);
	auto put = [&](int c, int t, C v)
	{
		out[((size_t) c * NT + t) * 2] = (double) v.real();
		out[((size_t) c * NT + t) * 2 + 1] = (double) v.imag();
	};
	for (int t = 0; t < NT; t++)
	{
		if (N2 < N)
		{
			const int D = N / N2, E2 = 16 / D;
			for (int cp = 0; cp < E2; cp++)
				put(cp, t, Hf(rev(16 * t + 2 * D * (cp >> 1) + ((cp & 1) ? 2 * D - 1 : 0), ln)));
		}
		else if (N2 == 2 * N)
		{
			for (int c = 0; c < 8; c++)
			{
				const int k = rev(8 * t + c, ln);
				put(c, t, Hf(k) + Hf(k + N));
				put(c + 8, t, Hf(k) - Hf(k + N));
			}
		}
		else
			for (int c = 0; c < 16; c++) put(c, t, Hf(rev(16 * t + c, ln)));
	}
	return out;
}

// Polyphase 3x form (ConvGeom::p3; r8b_convp.h mode 19): the spectra of the filter's three polyphase components
// g_r[d] = h[3 d + r + fl2], d in [-a, b], each laid out circularly so that the block's first valid output comes out at
// circular position 0 -- gc_r[(d - b) mod N] = g_r[d] --, scaled by 1 / N (unnormalised transforms both ways), one
// complex value per backward position: hp3[((r * 16 + c) * NT + t)] = G_r[bitrev(16 t + c)] (cf. pair_constants_complex).
std::vector<double> pair_constants_poly3(const LpFilter& f, int fl2, int N, int a, int b)
{
	const int NT = N / 16, K = (int) f.taps.size();
	int ln = 0;
	while ((1 << ln) < N) ln++;
	auto rev = [](int v, int bits)
	{
		int r = 0;
		for (int q = 0; q < bits; q++)
			if (v & (1 << q)) r |= 1 << (bits - 1 - q);
		return r;
	};
	const std::vector<double> tw = make_twiddles(N); // exp(-2 pi i e / N)
	std::vector<double> out((size_t) 3 * 16 * NT * 2, 0.0);
	for (int r = 0; r < 3; r++)
	{
		// (the component's taps with their circular positions)
		std::vector<std::pair<int, double>> taps;
		for (int d = -a; d <= b; d++)
		{
			const long long i = 3LL * d + r + fl2;
			if (i >= 0 && i < K) taps.emplace_back(((d - b) % N + N) % N, f.taps[(size_t) i]);
		}
		std::vector<long double> Gr((size_t) N), Gi((size_t) N);
		for (int k = 0; k <= N / 2; k++)
		{
			long double re = 0.0L, im = 0.0L;
			for (const auto& t : taps)
			{
				const long long e = ((long long) k * t.first) & (N - 1);
				re += (long double) t.second * tw[(size_t) e * 2];
				im += (long double) t.second * tw[(size_t) e * 2 + 1];
			}
			Gr[(size_t) k] = re / N;
			Gi[(size_t) k] = im / N;
			if (k > 0 && k < N / 2)
			{
				Gr[(size_t) (N - k)] = re / N;
				Gi[(size_t) (N - k)] = -im / N;
			}
		}
		for (int t = 0; t < NT; t++)
			for (int c = 0; c < 16; c++)
			{
				const int k = rev(16 * t + c, ln);
				const size_t o = (((size_t) r * 16 + c) * NT + t) * 2;
				out[o] = (double) Gr[(size_t) k];
				out[o + 1] = (double) Gi[(size_t) k];
			}
	}
	return out;
}

std::vector<double> pair_twiddles(const std::vector<double>& tw, int tw_len, int n_in, int n_out)
{
	// rows of NT entries (r8b_convp.h ptw_fetch): 6 per slot; slots 0-2 forward passes (radix e1); then,
	// 1:1 / 2x up: 3 the backward pass with sub-length 256, 4 + m butterfly m of the last backward pass
	// (sub-length n_out); decimating: 2 + I backward pass I (radix e2, sub-length rmb e2^I)
	const int NT = std::max(n_in, n_out) / 16, e1 = n_in / NT, e2 = n_out / NT;
	const bool dec = n_out < n_in || n_out > 4096; // (the mirrored backward side: decimating, or 8192 points)
	int npost = 0, rmb = 1;
	if (dec)
	{
		int ln2 = 0, eb2 = 0;
		while ((1 << ln2) < n_out) ln2++;
		while ((1 << eb2) < e2) eb2++;
		npost = (ln2 - 1) / eb2;
		rmb = 1 << (ln2 - npost * eb2);
	}
	const int r2 = n_out >= 256 ? n_out / 256 : n_out / 16;
	const int nb2 = r2 > 1 ? 16 / r2 : 0;
	const int nslots = dec ? 3 + npost : 4 + nb2;
	std::vector<double> out((size_t) nslots * 6 * NT * 2, 0.0);
	static const int mult[6] = { 1, 2, 3, 4, 8, 12 };
	for (int slot = 0; slot < nslots; slot++)
	{
		int n, jmod, joff = 0;
		if (slot < 3)
		{
			n = n_in;
			for (int i = 0; i < slot; i++) n /= e1;
			jmod = n / e1; // butterflies per sub-transform
			if (n < 2 * e1) continue;
		}
		else if (dec)
		{
			n = rmb;
			for (int i = 0; i < slot - 2; i++) n *= e2;
			jmod = n / e2;
		}
		else if (slot == 3)
		{
			if (n_out < 256) continue;
			n = 256; jmod = 16;
		}
		else { n = n_out; jmod = n_out; joff = NT * (slot - 4); }
		for (int t = 0; t < NT; t++)
			for (int c = 0; c < 6; c++)
			{
				const long long j = (t % jmod) + joff;
				const long long e = (long long) (tw_len / n) * j * mult[c];
				const size_t o = (((size_t) slot * 6 + c) * NT + t) * 2, i = (size_t) (e % tw_len) * 2;
				out[o] = tw[i];
				out[o + 1] = tw[i + 1];
			}
	}
	return out;
}

// LDS of the fast path: one padded complex array of n elements (r8b_convx.h, convx_lds_doubles)
static size_t convx_work_bytes(int n) { return (size_t) 2 * (n + (n >> 4)) * sizeof(double); }

// LDS of the generic convolver kernel: forward and backward arrays side by side; transforms of
// equal length can share one array (each spectral slot rewrites exactly the two bins it read)
static bool generic_conv_two_arrays(const ConvGeom& g)
{
	return (size_t) (g.n_in + g.n_out) * sizeof(double) <= 160 * 1024;
}
static bool generic_conv_fits(const ConvGeom& g)
{
	return generic_conv_two_arrays(g) ||
		(g.n_in == g.n_out && (size_t) g.n_in * sizeof(double) <= 160 * 1024);
}
// ... blocks of the reference's own length where the plan keeps them although the forward array does not fit (2^k
// decimation in the spectrum on 32768-point blocks, r8b_plan.cpp): the forward transform in two halves through LDS
// (k_conv_big)
static bool generic_conv_big(const ConvGeom& g)
{
	return !generic_conv_fits(g) && g.down_pow2 && g.down > 1 && g.n_in <= 32768 &&
		(size_t) g.n_out * sizeof(double) <= 128 * 1024;
}

std::vector<int> plan_radices(int N, int max_radix)
{
	std::vector<int> r;
	if (max_radix < 2) max_radix = 2;
	if (max_radix > 16) max_radix = 16;
	int bits = 0;
	while ((1 << bits) < N) bits++;
	int mb = 0;
	while ((2 << mb) <= max_radix) mb++;
	// spread the bits as evenly as possible over the fewest passes
	const int np = (bits + mb - 1) / mb;
	for (int i = 0; i < np; i++)
	{
		const int b = (bits + np - 1 - i) / np;
		r.push_back(1 << b);
	}
	return r;
}

// LDS row pitch of the tiled polynomial interpolator (doubles).  Lanes 0..15 and 16..31 of an LDS lane group
// read the windows of 16 consecutive outputs (input offsets floor(o * step + phase)) in two ADJACENT channel
// rows; banks repeat every 32 doubles, so the pitch's residue mod 32 decides whether the two rows' slots
// interleave (step 2: any odd residue) or collide.  Chosen by counting, for the step at hand.
static int poly_row_pitch(int span_max, double step)
{
	if (span_max <= 0) return 0;
	int best = 1;
	long best_cost = -1;
	for (int P = 0; P < 32; P++)
	{
		long cost = 0;
		for (int ph = 0; ph < 8; ph++)
		{
			int cnt[32] = { 0 };
			for (int o = 0; o < 16; o++)
			{
				const int xo = (int) std::floor(o * step + ph / 8.0);
				cnt[xo & 31]++;
				cnt[(xo + P) & 31]++;
			}
			int mx = 0;
			for (int i = 0; i < 32; i++) mx = std::max(mx, cnt[i]);
			cost += mx;
		}
		if (best_cost < 0 || cost < best_cost)
		{
			best_cost = cost;
			best = P;
		}
	}
	int pitch = span_max;
	while ((pitch & 31) != best) pitch++;
	return pitch;
}

static long long pow2_at_least(long long v)
{
	long long p = 1;
	while (p < v) p <<= 1;
	return p;
}

Engine::Engine(const std::vector<StageDesc>& descs, int maxin, int nch, int device)
	: nch_(nch), nchw_(nch), device_(dev_resolve(device))
{
	if (nch < 1) throw std::runtime_error("channel count must be >= 1");
	if (nch > 65535) throw std::runtime_error("channel count must be <= 65535 per batch object "
		"(grid y dimension); split larger batches");
	if (maxin < 1) throw std::runtime_error("MaxInLen must be >= 1");
	DevGuard guard(device_);
	plan_.init(descs, maxin);
	opt_["conv_radix"] = 8;
	opt_["conv_threads"] = 256;
	opt_["whole_tile"] = 4096;
	opt_["hb_tile"] = 1024;
	opt_["fuse_hbd"] = 2;    // runs of half-band decimators as one kernel: 0 never, 1 always, 2 by batch size
	opt_["hbd_span"] = 2048; // first-stage input samples per workgroup of the decimating cascade
	opt_["hbc_tile"] = 0; // last-stage outputs per workgroup of the half-band cascade (0: by batch)
	opt_["timing"] = 0;
	opt_["fast_conv"] = 1; // compile-time-sized convolver kernel when the geometry allows
	opt_["fuse"] = 1;      // ... with the whole-step interpolator behind it fused in
	opt_["fuse_hb"] = 1;   // runs of half-band up-samplers as one kernel
	opt_["poly_tiled"] = 1; // polynomial interpolator: 16 channels share each coefficient fetch
	// convolver + polynomial interpolator walked in channel groups (1: 96 MB between them; n > 1: n KB; 0: off).  Off:
	// measured on MI355X (profiles/r03_poly_groups.txt) the interpolator gains 2 % from reading the stream out of the
	// Infinity Cache and the convolver loses 8 % to its smaller launches (44100 -> 44101 x 1024 ch: 0.285 vs 0.280 ms)
	opt_["poly_groups"] = 0;
	// two channels per workgroup as one complex transform (r8b_convp.h) where the geometry allows
	opt_["pair_conv"] = 1;
	// ... with two adjacent phases per thread in the fused interpolator when it up-samples (In <= Out:
	// half the LDS reads per output, nearly all lanes busy)
	opt_["pair_two"] = 1;
	opt_["fuse_latency"] = 1; // chains with a fractional latency (minimum phase): convolver + interpolator in one launch too
	opt_["pair_split"] = 1;
	opt_["pair_solo"] = 1; // 16384-point 1:1 blocks on the pair kernel's one-channel form (else k_convx) // 8192 -> 16384-point 2x up-sampling blocks on the pair kernel's split form (else k_convx)
	opt_["align_groups"] = 1; // ... with whole output groups per block (launch_fused)
	opt_["fold_tail"] = 1; // fast convolver at stage 0 keeps the input history itself
	// the call's last block of a fused pair at the end of the chain is computed once: what it holds beyond the call
	// is parked for the next one (launch_fused)
	opt_["park"] = 1;
	// fused two-phase pair kernel in its walk form (r8b_convp.h convp_walk): a workgroup per channel pair takes the call's
	// blocks one after the other (0: a workgroup per block, as before round 5)
	// 16384-point 1:1 blocks (one-channel form of the pair kernel) with the whole-step interpolator behind them fused in
	// (kernel mode 18; 0: the interpolator as a launch of its own, as before round 5)
	opt_["solo_fuse"] = 1;
	// 3x up-sampling convolvers in the polyphase form -- one forward transform of the INPUT samples, three backward ones,
	// no stuffed zeros transformed (r8b_convp.h mode 19, ConvGeom::p3); 0: the zero-stuffing block, as before round 5
	opt_["up3_poly"] = 1;
	// eight elements per thread (r8b_convq.h): the 2048 -> 4096-point convolver-only block pair on 512 threads (four waves
	// per SIMD instead of two); same blocks, same state, results differ from the 256-thread form by rounding
	opt_["quad"] = 0;
	// half-array form of that block pair (r8b_convp.h cp_ha_*, kernel mode 21): the backward side's two exchanges move the
	// real parts, then the imaginary parts through an array of DOUBLES -- 32 KB, four workgroups per CU, no pass added.
	// Measured on MI355X (profiles/r06_experiments.txt item 12): 44100 -> 88200 at 1024 channels 0.1429 -> 0.1277 ms per
	// call (kernel 0.138 -> 0.118), at 256 channels -9 %, at 4096 -13 %; launches that do not fill the chip twice gain
	// nothing (64 channels: +0.5 %).  1: objects whose largest call holds at least 512 workgroups of the stage -- channel
	// pairs x blocks, Engine::half_worth -- (decided per OBJECT, never per call: the two
	// forms do the same arithmetic on the same values -- bitwise equal under host emulation -- but the device compiler
	// contracts multiply-adds differently in the two kernels, so on the GPU they agree to rounding, 4e-17 RMS, and an
	// object must stay with one of them to remain bitwise chunk invariant); 2: every object; 0: the 64 KB form
	opt_["half"] = 1;
	// ... and of the fused two-phase block pair 2048 -> 4096 points + whole-step interpolator (kernel modes 23 / 25: the
	// array is the interpolator's run, 52 KB with flag words and twiddle table, three workgroups per CU; taken in place of
	// modes 4 / 5 and of the walk form).  Measured on cfg2 (profiles/r06_experiments.txt item 13): -1.3 ... -2.1 % per call
	// against the walk form, -4.2 % against a workgroup per block; kernel events level with the walk form.  Values as for
	// "half"
	opt_["half_fused"] = 1;
	// a half-band decimator in front of a 4096 -> 2048-point decimating convolver taken in the convolver's load (kernel mode
	// 20: one launch, the decimator's stream never leaves LDS).  Off: measured on MI355X the fused launch takes 92.6 us
	// + a 19 us history copy against 52.5 + 42.6 us for the two launches (176400 -> 44100, 1024 ch x 16384) -- a block
	// cannot start before its 133 KB of raw samples have arrived and all workgroups ask at once: 38 000 of a block's
	// 66 000 cycles are the two staging rounds at 4.2 TB/s (profiles/r06_experiments.txt item 10); the raw-domain
	// history a call has to leave (avg 5 300 samples per channel) is 4x the convolver-domain one besides
	opt_["fuse_hbconv"] = 0;
	opt_["walk"] = 1;      // (0: a workgroup per block, as before round 5; 2: whatever the batch size -- tests)
	opt_["walk_len"] = 0;  // blocks per workgroup of the walk form (0: the launch's whole run of blocks)
	stat_["conv_blocks"] = 0;
	stat_["walk_blocks"] = 0; // blocks of the fused pair kernel's launches that ran on the walk body (per channel, like conv_blocks)
	stat_["tail_launches"] = 0; // history copies that needed a launch of their own (k_tail)
	stat_["park_calls"] = 0;
	stat_["park_only_calls"] = 0;
	stat_["pcm_staged_sides"] = 0; // planar PCM sides that went through the staging rows (r8b_capi.cpp)
	// a constructor that throws half way must not leak what it has already put on the device
	try
	{
		dev_.resize(plan_.stages.size());
		for (size_t s = 0; s < plan_.stages.size(); s++)
		{
			const StagePlan& sp = plan_.stages[s];
			StageDev& d = dev_[s];
			// (a decimator that option fuse_hbconv may take into the convolver behind it: the larger history of the two forms)
			const long long hist = std::max(stage_history(s), hbconv_possible(s) ? hbconv_history(s) : 0LL);
			// (+ one block of the convolver in front of it: the block that holds a call's last output is written whole,
			// ahead of what the call owes -- launch_stage, conv_once)
			// (the same behind a fused convolver + whole-step interpolator: a block's interpolated outputs -- launch_fused)
			// (the larger of the blocks the convolver may run on: the plan's, or its polyphase 3x block -- ConvGeom::p3)
			long long ahead = s > 0 && plan_.stages[s - 1].desc.kind == kConv ?
				std::max(plan_.stages[s - 1].cg.in_len, plan_.stages[s - 1].cg.p3 ? 3 * plan_.stages[s - 1].cg.p3_m : 0) /
					plan_.stages[s - 1].cg.down + 2 : 0;
			if (s > 1 && plan_.stages[s - 1].desc.kind == kFrac && plan_.stages[s - 1].whole &&
				plan_.stages[s - 2].desc.kind == kConv)
				ahead = (long long) plan_.stages[s - 2].cg.in_len * plan_.stages[s - 1].out_step / plan_.stages[s - 1].in_step +
					plan_.stages[s - 1].out_step + 2;
			d.ring_size = pow2_at_least(s == 0 ? hist : hist + plan_.stage_max_in[s] + ahead);
			// rings are allocated on first use (ensure_ring): the ring between two fused stages is
			// never touched and would be the largest allocation (cfg2: 512 MB)
			if (sp.desc.kind == kConv)
			{
				const ConvGeom& g = sp.cg;
				if (g.n_in < 32 || g.n_out < 32)
					throw std::runtime_error("block convolver transform too short");
				if (g.p3)
				{
					// polyphase 3x form: the three component spectra and the twiddles of its own geometry (beside the tables of
					// the zero-stuffing block, which option up3_poly = 0 falls back to)
					const std::vector<double> h3 = pair_constants_poly3(*sp.lp, g.fl2, g.p3_n, g.p3_a, g.p3_b);
					d.hp3 = (cd*) dev_alloc(h3.size() * sizeof(double));
					dev_upload(d.hp3, h3.data(), h3.size() * sizeof(double));
					const std::vector<double> tw3 = make_twiddles(g.p3_n);
					const std::vector<double> pt3 = pair_twiddles(tw3, g.p3_n, g.p3_n, g.p3_n);
					d.ptw3 = (cd*) dev_alloc(pt3.size() * sizeof(double));
					dev_upload(d.ptw3, pt3.data(), pt3.size() * sizeof(double));
				}
				// the generic kernel keeps both transforms' arrays in LDS, the fast path works in place
				const bool m3 = convx_mode3_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2);
				const bool fast_ok = (m3 || convx_geometry_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2)) &&
					convx_work_bytes(std::max(g.n_in, g.n_out) / 2) <= 160 * 1024;
				if (!generic_conv_fits(g) && !fast_ok && !generic_conv_big(g))
					throw std::runtime_error("low-pass filter too long for the LDS-resident "
						"block convolver (transition band too narrow)");
				if (g.complex_h)
				{
					// minimum phase (or an alignment moved by inherited latency): complex spectrum for the
					// generic kernel ...
					// ... (16384 points 1:1 in place; 2x up-sampling or decimating at that length: neither array pair fits)
					const bool split_cx = convp_split_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2);
					const bool solo_cx = !split_cx && g.n_in == g.n_out &&
						convp_solo_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2, g.in_len);
					const bool down_cx = g.down == 2 &&
						convp_solo_down_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2, g.in_len);
					if (!generic_conv_fits(g) && !split_cx && !solo_cx && !down_cx && !generic_conv_big(g))
						throw std::runtime_error("minimum-phase filter too long for the generic block convolver");
					const std::vector<double> hc = kernel_spectrum_complex(*sp.lp, g.bl2, g.fl2, 1.0 / g.bl2);
					d.Hc = (cd*) dev_alloc(hc.size() * sizeof(double));
					dev_upload(d.Hc, hc.data(), hc.size() * sizeof(double));
					const std::vector<double> tw = make_twiddles(g.bl2);
					d.tw_len = g.bl2;
					d.tw = (cd*) dev_alloc(tw.size() * sizeof(double));
					dev_upload(d.tw, tw.data(), tw.size() * sizeof(double));
					if (split_cx || solo_cx || down_cx)
					{
						// ... the long-block forms of the pair kernel (modes 12 ... 15)
						const int n = split_cx ? g.n_in : g.n_in / 2;
						const std::vector<double> hp = split_cx ? pair_constants_split_complex(hc, n) :
							(down_cx ? pair_constants_solo_down_complex(hc, n) : pair_constants_solo_complex(hc, n));
						d.hp = (cd*) dev_alloc(hp.size() * sizeof(double));
						dev_upload(d.hp, hp.data(), hp.size() * sizeof(double));
						const std::vector<double> pt = pair_twiddles(tw, g.bl2, n, down_cx ? n / 2 : n);
						d.ptw = (cd*) dev_alloc(pt.size() * sizeof(double));
						dev_upload(d.ptw, pt.data(), pt.size() * sizeof(double));
					}
					else if (convp_geometry_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2) ||
						convp_mode3_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2))
					{
						// ... and the pair kernel with one complex multiplication per bin
						const std::vector<double> hp = pair_constants_complex(hc, g.n_in, g.n_out);
						d.hp = (cd*) dev_alloc(hp.size() * sizeof(double));
						dev_upload(d.hp, hp.data(), hp.size() * sizeof(double));
						const std::vector<double> pt = pair_twiddles(tw, g.bl2, g.n_in, g.n_out);
						d.ptw = (cd*) dev_alloc(pt.size() * sizeof(double));
						dev_upload(d.ptw, pt.data(), pt.size() * sizeof(double));
					}
					continue;
				}
				const std::vector<double> H = kernel_spectrum(*sp.lp, g.bl2, 1.0 / g.bl2);
				d.H = (double*) dev_alloc(H.size() * sizeof(double));
				dev_upload(d.H, H.data(), H.size() * sizeof(double));
				const std::vector<double> tw = make_twiddles(g.bl2);
				d.tw_len = g.bl2;
				d.tw = (cd*) dev_alloc(tw.size() * sizeof(double));
				dev_upload(d.tw, tw.data(), tw.size() * sizeof(double));
				if (m3 || convx_geometry_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2))
				{
					// (3x zero stuffing / 3x strided decimation are 1:1 for the transforms)
					const int eup = g.up_pow2 ? g.up : 1, edown = g.down_pow2 ? g.down : 1;
					const std::vector<double> sc = edown > 1 ?
						spectral_constants_down(H, tw, g.bl2, g.n_in, edown) :
						spectral_constants(H, tw, g.bl2, g.n_in, eup);
					d.spec = (cd*) dev_alloc(sc.size() * sizeof(double));
					dev_upload(d.spec, sc.data(), sc.size() * sizeof(double));
					if (edown == 1)
					{
						const std::vector<double> s2 = spectral_constants_by_position(sc, g.n_in, eup);
						d.spec2 = (cd*) dev_alloc(s2.size() * sizeof(double));
						dev_upload(d.spec2, s2.data(), s2.size() * sizeof(double));
					}
				}
				if (convp_solo_down_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2, g.in_len))
				{
					// one-channel form decimating by 2: the passes of the 8192 -> 4096-point geometry
					const std::vector<double> hp = pair_constants_solo_down(H, g.n_in / 2, g.down);
					d.hp = (cd*) dev_alloc(hp.size() * sizeof(double));
					dev_upload(d.hp, hp.data(), hp.size() * sizeof(double));
					const std::vector<double> pt = pair_twiddles(tw, g.bl2, g.n_in / 2, g.n_out / 2);
					d.ptw = (cd*) dev_alloc(pt.size() * sizeof(double));
					dev_upload(d.ptw, pt.data(), pt.size() * sizeof(double));
				}
				else if (convp_solo_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2, g.in_len))
				{
					// one-channel form: constants per forward position, the passes of the 8192-point 1:1 geometry
					const std::vector<double> hp = pair_constants_solo(H, g.n_in / 2);
					d.hp = (cd*) dev_alloc(hp.size() * sizeof(double));
					dev_upload(d.hp, hp.data(), hp.size() * sizeof(double));
					const std::vector<double> pt = pair_twiddles(tw, g.bl2, g.n_in / 2, g.n_in / 2);
					d.ptw = (cd*) dev_alloc(pt.size() * sizeof(double));
					dev_upload(d.ptw, pt.data(), pt.size() * sizeof(double));
				}
				else if (convp_split_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2))
				{
					// split 2x up-sampling form: constants per forward position, the passes of the 1:1 geometry
					const std::vector<double> hp = pair_constants_split(H, g.n_in);
					d.hp = (cd*) dev_alloc(hp.size() * sizeof(double));
					dev_upload(d.hp, hp.data(), hp.size() * sizeof(double));
					const std::vector<double> pt = pair_twiddles(tw, g.bl2, g.n_in, g.n_in);
					d.ptw = (cd*) dev_alloc(pt.size() * sizeof(double));
					dev_upload(d.ptw, pt.data(), pt.size() * sizeof(double));
				}
				else if (convp_geometry_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2) ||
					convp_mode3_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2))
				{
					const std::vector<double> hp = pair_constants(H, g.n_in, g.n_out);
					d.hp = (cd*) dev_alloc(hp.size() * sizeof(double));
					dev_upload(d.hp, hp.data(), hp.size() * sizeof(double));
					const std::vector<double> pt = pair_twiddles(tw, g.bl2, g.n_in, g.n_out);
					d.ptw = (cd*) dev_alloc(pt.size() * sizeof(double));
					dev_upload(d.ptw, pt.data(), pt.size() * sizeof(double));
				}
			}
			else if (sp.desc.kind == kFrac)
			{
				const std::vector<double>& t = sp.bank->table;
				d.table = (double*) dev_alloc(t.size() * sizeof(double));
				dev_upload(d.table, t.data(), t.size() * sizeof(double));
				if (sp.whole)
				{
					std::vector<double> w((size_t) sp.flen * sp.out_step);
					for (int r = 0; r < sp.out_step; r++)
					{
						const int ph = (int) (((long long) r * sp.in_step) % sp.out_step);
						for (int i = 0; i < sp.flen; i++)
							w[(size_t) i * sp.out_step + r] = t[(size_t) ph * sp.flen + i];
					}
					d.wtab = (double*) dev_alloc(w.size() * sizeof(double));
					dev_upload(d.wtab, w.data(), w.size() * sizeof(double));
				}
			}
		}
		plan_transforms();
		for (size_t s = 0; s + 1 < plan_.stages.size(); s++)
			if (fuse_with_next(s)) prepare_two_phase(s);
	}
	catch (...)
	{
		release();
		throw;
	}
}

// Tables of the pair kernel's two-phases-per-thread interpolator (r8b_convp.h MODE 4) for the fused
// pair (convolver s, whole-step interpolator s+1); needs In <= Out and at most 24 taps.
// whether a whole-step interpolator has two-phase tables at all (prepare_two_phase): at most 24 taps, 2 ... 510 phases,
// the window starts of a phase pair at most three samples apart; *maxdl_out: that distance
static bool two_phase_possible(const StagePlan& w, int* maxdl_out)
{
	const int In = w.in_step, Out = w.out_step;
	if (w.flen > 24 || Out < 2 || Out > 510) return false;
	const int NP = (Out + 1) / 2, nsg = (NP + 15) / 16;
	if (16 / nsg < 1) return false;
	int maxdl = 0;
	for (int q = 0; 2 * q + 1 < Out; q++)
		maxdl = std::max(maxdl, (int) ((long long) (2 * q + 1) * In / Out) - (int) ((long long) (2 * q) * In / Out));
	if (maxdl_out) *maxdl_out = maxdl;
	return maxdl <= 3;
}

void Engine::prepare_two_phase(size_t s)
{
	const StagePlan& w = plan_.stages[s + 1];
	StageDev& d = dev_[s + 1];
	const int In = w.in_step, Out = w.out_step;
	int maxdl = 0;
	if (!two_phase_possible(w, &maxdl)) return;
	const int NP = (Out + 1) / 2;           // phase pairs
	const int nsg = (NP + 15) / 16;         // 16-lane LDS service groups per set
	const int nsets = 16 / nsg;             // a workgroup has 16 service groups
	auto r_of = [&](int ph) { return (int) ((long long) ph * In / Out); };
	// rows of T2 entries: the taps plus the largest distance between the window starts of a pair
	const int T2 = maxdl <= 1 ? 25 : 27;
	// Phase pairs go to lanes in QUADS: the four lanes of an aligned lane quad own four consecutive phase pairs of one
	// group set, i.e. store 64 consecutive bytes of a channel's output per group (one L2 write request per quad;
	// assigned pair by pair, the 16-byte pieces of a store instruction were scattered over the whole group: 19 bytes
	// per request measured).  A workgroup has 16 LDS service groups of four lane quads each (the 16 lanes LDS serves
	// in one cycle of a 16-byte read), and a service group takes as many cycles as its most crowded 16-byte bank
	// group: ONE pair of windows starting in the same bank group (start mod 16) doubles the time of all its reads.
	// The (data quad, set) items -- NQ x nsets <= 64 -- are therefore dealt to the 64 (service group, lane quad)
	// slots such that as few service groups as possible hold a clash.  Items of DIFFERENT sets may share a service
	// group (a set shifts the window starts by In mod 16), and slots may stay empty: with both freedoms cfg2 comes to
	// 2 clashing service groups of 16, where set-by-set dealing left one clash in every group (7.5 LDS cycles per read
	// instead of 4; counters: bank conflicts 30 % of the LDS cycles).  Deterministic annealing over slot swaps (fixed
	// seed: the same tables for the same ratio, always).
	const int NQ = (NP + 3) / 4, NI = NQ * nsets;
	std::vector<int> slot_item(64, -1); // [service group * 4 + rank] -> item = set * NQ + data quad
	// (the table depends on the ratio alone: searched once per process and ratio)
	// (bounded like the designer's caches -- kLaneDealCacheMax ratios, most recently used first; an entry is copied out,
	// so dropping the oldest one costs a host that comes back to its ratio one more search, nothing else)
	static std::mutex cache_mutex;
	static std::list<std::pair<std::pair<int, int>, std::vector<int>>> cache;
	bool cached = false;
	{
		std::lock_guard<std::mutex> lock(cache_mutex);
		for (auto it = cache.begin(); it != cache.end(); ++it)
			if (it->first == std::make_pair(In, Out))
			{
				cache.splice(cache.begin(), cache, it);
				slot_item = cache.front().second;
				cached = true;
				break;
			}
	}
	if (!cached)
	{
		auto res_of = [&](int item, int j) // bank group of the window start of pair j of the item (-1: no such pair)
		{
			const int dq = item % NQ, set = item / NQ, q = 4 * dq + j;
			return q < NP ? (int) ((r_of(2 * q) + (long long) In * set) & 15) : -1;
		};
		auto sg_cost = [&](const int* sl) // clashes of one service group
		{
			int cnt[16] = { 0 }, c = 0;
			for (int k = 0; k < 4; k++)
				if (sl[k] >= 0)
					for (int j = 0; j < 4; j++)
					{
						const int r = res_of(sl[k], j);
						if (r >= 0 && cnt[r]++) c++;
					}
			return c;
		};
		// start: set by set, data quads in order
		for (int i = 0; i < NI; i++)
		{
			const int set = i / NQ, dq = i % NQ;
			slot_item[(size_t) ((set * nsg + dq / 4) * 4 + dq % 4)] = i;
		}
		auto total = [&]()
		{
			int c = 0;
			for (int g = 0; g < 16; g++) c += sg_cost(&slot_item[(size_t) g * 4]);
			return c;
		};
		std::vector<int> best = slot_item;
		int cur = total(), best_cost = cur;
		unsigned long long rng = 0x9E3779B97F4A7C15ull;
		auto next = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (unsigned) (rng >> 33); };
		double temp = 1.0;
		for (int it = 0; it < 250000 && best_cost > 0; it++)
		{
			const int a = (int) (next() % 64u), b = (int) (next() % 64u);
			if (a / 4 == b / 4 || (slot_item[(size_t) a] < 0 && slot_item[(size_t) b] < 0)) continue;
			int* ga = &slot_item[(size_t) (a / 4) * 4];
			int* gb = &slot_item[(size_t) (b / 4) * 4];
			const int before = sg_cost(ga) + sg_cost(gb);
			std::swap(slot_item[(size_t) a], slot_item[(size_t) b]);
			const int after = sg_cost(ga) + sg_cost(gb);
			bool keep = after <= before;
			if (!keep)
			{
				// (accept a worse state with probability exp(-(after - before) / temp))
				const double u = (next() & 0xffffff) / 16777216.0;
				keep = u < std::exp(-(after - before) / temp);
			}
			if (keep) cur += after - before;
			else std::swap(slot_item[(size_t) a], slot_item[(size_t) b]);
			if (cur < best_cost)
			{
				best_cost = cur;
				best = slot_item;
			}
			temp = std::max(0.05, temp * 0.99998);
		}
		slot_item = best;
		{
			std::lock_guard<std::mutex> lock(cache_mutex);
			bool have = false; // (another thread may have searched the same ratio meanwhile: same table, one entry)
			for (const auto& e : cache) have = have || e.first == std::make_pair(In, Out);
			if (!have)
			{
				int dropped = 0;
				while (cache.size() >= (size_t) kLaneDealCacheMax)
				{
					cache.pop_back();
					dropped++;
				}
				cache.emplace_front(std::make_pair(In, Out), slot_item);
				lane_deal_cache_count(1 - dropped, nullptr);
			}
		}
		if (std::getenv("R8B_DEBUG_TWO")) fprintf(stderr, "two-phase tables: %d phase pairs in %d quads x %d sets, %d bank clashes over the 16 service groups\n", NP, NQ, nsets, best_cost);
	}
	// lanes LDS serves together for 16-byte reads (MI355X_MICROARCH.md, LDS): within each half of a
	// wave the quads {0, 3, 5, 6} and {1, 2, 4, 7}
	std::vector<int> pt(256, -1);
	const int ctp = (NP + 3) & ~3;
	std::vector<double> ct((size_t) 2 * T2 * ctp, 0.0);
	const std::vector<double>& T = w.bank->table;
	for (int t = 0; t < 256; t++)
	{
		const int wave = t >> 6, lane = t & 63, half = lane >> 5, quad = (lane & 31) >> 2;
		static const int cls_of[8] = { 0, 1, 1, 0, 1, 0, 0, 1 }, rank_of[8] = { 0, 0, 1, 1, 2, 2, 3, 3 };
		const int sg = 4 * wave + 2 * half + cls_of[quad];
		const int item = slot_item[(size_t) (sg * 4 + rank_of[quad])];
		if (item < 0) continue;
		const int set = item / NQ;
		const int q = 4 * (item % NQ) + (lane & 3);
		if (q >= NP) continue;
		pt[(size_t) t] = q | (set << 8) | (r_of(2 * q) << 12);
		const int p0 = 2 * q, p1 = 2 * q + 1;
		const int row0 = (int) (((long long) p0 * In) % Out);
		// value v (0 .. 2 T2 - 1: the T2 taps of phase p0, then of p1 shifted by its window offset) of phase PAIR q
		// sits in pair v / 2: ct[((v / 2) * ctp + q) * 2 + v % 2], ctp = NP rounded up to whole quads.  Indexed by the
		// pair, not by the thread: the lanes of the nsets sets that own the same pair read the same entries, and a
		// workgroup pulls every row through L2 once instead of nsets times (cfg2: 34 KB instead of 102 KB per block).
		auto put = [&](int v, double x) { ct[((size_t) (v / 2) * ctp + q) * 2 + (v & 1)] = x; };
		for (int i = 0; i < w.flen; i++) put(i, T[(size_t) row0 * w.flen + i]);
		if (p1 < Out)
		{
			const int row1 = (int) (((long long) p1 * In) % Out), dl = r_of(p1) - r_of(p0);
			for (int i = 0; i < w.flen; i++) put(T2 + i + dl, T[(size_t) row1 * w.flen + i]);
		}
	}
	d.ptab = (int*) dev_alloc(pt.size() * sizeof(int));
	dev_upload(d.ptab, pt.data(), pt.size() * sizeof(int));
	d.ctab = (double*) dev_alloc(ct.size() * sizeof(double));
	dev_upload(d.ctab, ct.data(), ct.size() * sizeof(double));
	d.nsets = nsets;
	d.taps2 = T2;
}

// Is a half-array form (r8b_convp.h cp_ha_*) worth taking for convolver stage s?  The forms pay where a launch holds more
// workgroups than the chip has slots for the full-array kernel -- two per CU, 512 --: channel pairs x blocks of the object's
// LARGEST call.  A constant of the object (its channel count, its MaxInLen, the stage's block), never of a call: on the
// device the two forms of a kernel agree to rounding only, and a stream stays with one.  Measured (r06_experiments.txt
// item 12): 1024 ch x 16384 -10.6 %, 256 ch -8.9 %, 64 ch x 16384 (384 workgroups) +0.6 %, 64 ch x 1024 (32) +0.5 %.
bool Engine::half_worth(size_t s) const
{
	const ConvGeom& g = plan_.stages[s].cg;
	const long long pairs = ((long long) nch_ + 1) / 2;
	const long long per_block = std::max(1, g.in_len / std::max(1, g.up)); // stage input samples one block brings
	const long long blocks = ((long long) plan_.stage_max_in[s] + per_block - 1) / per_block;
	return pairs * blocks >= 512;
}

bool Engine::use_pair_two(size_t s, int* run_off) const
{
	const StagePlan& c = plan_.stages[s];
	const StagePlan& w = plan_.stages[s + 1];
	const StageDev& dw = dev_[s + 1];
	if (!opt_.at("pair_two") || !use_pair_fused(c.cg) || dw.ptab == nullptr) return false;
	const int off = (w.in_step + 16 + 15) / 16 * 16;
	if (off + c.cg.in_len + w.in_step + 32 + 16 > c.cg.n_out) return false;
	if (run_off) *run_off = off;
	return true;
}

// Does stage s keep a park buffer for the outputs its call's last block holds beyond the call?  A fused pair
// (convolver s, whole-step interpolator s + 1) in the two-phase pair form, or a pair-kernel convolver on its own, AT THE
// END OF THE CHAIN: the caller's rows are the one destination that cannot take outputs ahead of their call.  (In the
// middle of a chain the same block simply writes ahead into the next stage's ring: conv_once.)  A constant of the
// object and its options.
bool Engine::stage_parks(size_t s) const
{
	if (!opt_.at("park") || s >= plan_.stages.size() || plan_.stages[s].desc.kind != kConv) return false;
	if (s + 2 == plan_.stages.size() && fuse_with_next(s))
		return use_pair_two(s, nullptr) || !use_pair_fused(plan_.stages[s].cg); // (the latter: an output ring)
	if (s + 1 != plan_.stages.size()) return false;
	const int path = conv_path(eff_geom(s));
	// (the one-channel fast path at the end of a chain keeps an output ring in the same buffer: launch_stage)
	return path == kPathPair || path == kPathPair3 || path == kPathPairP3 || path == kPathConvx || path == kPathConvx3;
}

// How an unfused pair-kernel convolver treats the block that holds a call's last output (launch_stage): 0 -- computed
// again by the next call (option park = 0); 2 -- computed once, the outputs beyond the call parked (end of the chain,
// fp64 rows of the caller); 3 -- computed once, written ahead into the next stage's ring.
int Engine::conv_once(size_t s, const DstView& dst) const
{
	if (!opt_.at("park")) return 0;
	if (s + 1 == plan_.stages.size()) return dst.mask == -1 && dst.fmt == kPcmF64 && stage_parks(s) ? 2 : 0;
	return dst.mask != -1 && dst.fmt == kPcmF64 ? 3 : 0;
}

// doubles per channel of a park buffer: what one block can hold, rounded up to whole 64-byte lines
long long Engine::park_row_len(size_t s) const
{
	return park_len_of(s, true);
}

// (end_of_chain = false: the bound on what a stage in the MIDDLE of a chain writes ahead into the next ring -- one
// block's outputs, whatever kernel runs it; load_state checks a blob's counters against it)
long long Engine::park_len_of(size_t s, bool end_of_chain) const
{
	long long n;
	if ((!end_of_chain || s + 2 == plan_.stages.size()) && fuse_with_next(s))
	{
		long long S = 0, off = 0;
		fused_blocking(s, &S, &off);
		const StagePlan& w = plan_.stages[s + 1];
		n = (S * w.out_step + w.in_step - 1) / w.in_step + 2;
		// (block 0 holds every output whose window ends inside its valid run [-fl2, in_len - fl2): more than a later
		// block's share when fl2 is small -- a minimum-phase filter's few samples)
		const ConvGeom& g = plan_.stages[s].cg;
		const long long e0 = (long long) g.in_len + off - g.fl2 - w.fl2 - fused_shift(s).d;
		if (e0 > 0) n = std::max(n, (e0 * w.out_step + w.in_step - 1) / w.in_step + 2);
		// (output ring of the one-channel fused kernel: a call's outputs plus one block's, a power of two)
		if (end_of_chain && !use_pair_fused(plan_.stages[s].cg) && !use_solo_fused(s))
			return pow2_at_least(plan_.max_out_len + n + 16);
	}
	else
	{
		const ConvGeom g = eff_geom(s);
		n = g.in_len / g.down + 2;
		const int path = conv_path(g);
		// (output ring of the one-channel fast path: a call's outputs plus one block's, a power of two)
		if (end_of_chain && (path == kPathConvx || path == kPathConvx3)) return pow2_at_least(plan_.max_out_len + n + 16);
	}
	return (n + 7) / 8 * 8 + 8;
}

void Engine::ensure_park(size_t s)
{
	StageDev& d = dev_[s];
	const long long len = park_row_len(s);
	if (d.park[0] != nullptr && d.park_stride == len) return;
	// (the row length follows the structural options -- pair_solo, pair_conv, align_groups ... --, which may change
	// between clear() and the next process(): buffers of another length are replaced, and they hold nothing then)
	if (d.park[0] != nullptr)
	{
		if (d.park_end > d.park_base) throw std::logic_error("park buffer resized while it holds outputs");
		dev_free(d.park[0]);
		dev_free(d.park[1]);
		d.park[0] = d.park[1] = nullptr;
		d.park_cur = 0;
	}
	d.park_stride = len;
	const size_t bytes = (size_t) d.park_stride * (size_t) nch_ * sizeof(double);
	d.park[0] = (double*) dev_alloc(bytes);
	d.park[1] = (double*) dev_alloc(bytes);
}

Engine::~Engine() { release(); }

void Engine::release()
{
	if (dev_.empty()) return;
	// (called from the destructor: a device that cannot be made current any more -- runtime unloaded at interpreter
	// exit, sticky error -- must not throw here; the frees below then fail silently)
	int prev = -1;
	try { prev = dev_swap(device_); } catch (...) { prev = -1; }
	struct Restore { int prev; ~Restore() { if (prev >= 0) dev_restore(prev); } } restore{prev};
	for (StageDev& d : dev_)
	{
		for (auto& pr : d.pending)
		{
			dev_event_destroy(pr.first);
			dev_event_destroy(pr.second);
		}
		for (void* e : d.free_events) dev_event_destroy(e);
		dev_free(d.ring);
		dev_free(d.ring_alt);
		dev_free(d.H);
		dev_free(d.Hc);
		dev_free(d.tw);
		dev_free(d.spec);
		dev_free(d.spec2);
		dev_free(d.hp);
		dev_free(d.ptw);
		dev_free(d.hp3);
		dev_free(d.ptw3);
		dev_free(d.table);
		dev_free(d.wtab);
		dev_free(d.ptab);
		dev_free(d.ctab);
		dev_free(d.park[0]);
		dev_free(d.park[1]);
		dev_free(d.work);
	}
	dev_.clear();
}

void Engine::plan_transforms()
{
	for (size_t s = 0; s < plan_.stages.size(); s++)
	{
		const StagePlan& sp = plan_.stages[s];
		if (sp.desc.kind != kConv) continue;
		const bool big = generic_conv_big(sp.cg);
		dev_[s].fwd_radix = plan_radices(sp.cg.n_in / 2, opt_["conv_radix"]);
		std::vector<int> inv = plan_radices(sp.cg.n_out / 2, big ? 16 : opt_["conv_radix"]);
		if (big)
		{
			// (k_conv_big: a radix-2 stage in the load, then the two sub-blocks of half the length on their own, on 512
			// threads -- one radix-16 butterfly each)
			dev_[s].fwd_radix = plan_radices(sp.cg.n_in / 4, 16);
			dev_[s].fwd_radix.insert(dev_[s].fwd_radix.begin(), 2);
		}
		// backward passes run with growing sub-transform length: smallest radix group first
		dev_[s].inv_radix.assign(inv.rbegin(), inv.rend());
	}
}

bool Engine::set_option(const std::string& name, int value)
{
	auto it = opt_.find(name);
	if (it == opt_.end()) return false;
	// Options that choose between fused and unfused kernels decide where a stage's history lives
	// (unfused stages keep it in rings the fused kernels never write): once a stream has started they
	// may only change after clear().
	static const char* const structural[] = { "fuse", "fuse_hb", "fuse_hbd", "fuse_hbconv", "fold_tail", "fast_conv",
		"pair_conv", "pair_two", "pair_split", "pair_solo", "align_groups", "park", "fuse_latency", "solo_fuse", "up3_poly",
		// (the half-array forms keep the state where it is, but round differently on the device: a stream stays with one)
		"half", "half_fused", "quad" };
	bool started = false;
	for (const StagePlan& sp : plan_.stages) started = started || sp.m != 0;
	for (const char* n : structural)
		if (started && name == n && it->second != value) return false;
	it->second = value;
	plan_transforms();
	return true;
}

long long Engine::stat(const std::string& name) const
{
	auto it = stat_.find(name);
	return it == stat_.end() ? -1 : it->second;
}

void Engine::ensure_ring(size_t s)
{
	StageDev& d = dev_[s];
	if (d.ring != nullptr) return;
	const size_t bytes = (size_t) d.ring_size * (size_t) nch_ * sizeof(double);
	d.ring = (double*) dev_alloc(bytes);
	if (s == 0) d.ring_alt = (double*) dev_alloc(bytes);
}

// backward-spectrum arrays of the long-block generic convolver (k_conv_big), one per workgroup of its launch: grown,
// never shrunk; growing waits for the device (the arrays may be in use by an earlier launch), which happens on an
// object's first calls only
void Engine::ensure_work(size_t s, int slots, void* stream)
{
	StageDev& d = dev_[s];
	if (d.work != nullptr && d.work_slots >= slots) return;
	if (d.work != nullptr)
	{
		dev_sync(stream); // (hipFree waits for the device besides)
		dev_free(d.work);
		d.work = nullptr;
	}
	d.work = (double*) dev_alloc((size_t) slots * (size_t) plan_.stages[s].cg.n_out * sizeof(double));
	d.work_slots = slots;
}

// a half-band launch takes the call's pending history copy with it (Engine::process)
void Engine::take_carried_tail(TailLaunch& T, int* carry)
{
	*carry = 0;
	T = carry_tail_;
	if (!carry_) return;
	carry_ = false;
	tail_done_ = true;
	if (carry_tail_.p1 > carry_tail_.p0) *carry = 1;
}

void* Engine::get_event(StageDev& d)
{
	if (!d.free_events.empty())
	{
		void* e = d.free_events.back();
		d.free_events.pop_back();
		return e;
	}
	return dev_event_create();
}

bool Engine::stage_timing(size_t stage, double* ms_sum, int* launches, std::string* kernel,
	long long* in_samples, long long* out_samples)
{
	if (stage >= dev_.size()) return false;
	StageDev& d = dev_[stage];
	for (auto& pr : d.pending)
	{
		d.ms_sum += dev_event_elapsed_ms(pr.first, pr.second);
		d.launches++;
		d.free_events.push_back(pr.first);
		d.free_events.push_back(pr.second);
	}
	d.pending.clear();
	if (ms_sum) *ms_sum = d.ms_sum;
	if (launches) *launches = d.launches;
	if (in_samples) *in_samples = d.t_in;
	if (out_samples) *out_samples = d.t_out;
	if (kernel)
	{
		const StagePlan& sp = plan_.stages[stage];
		switch (sp.desc.kind)
		{
		case kConv:
			*kernel = fuse_with_next(stage) ?
				(use_pair_fused(sp.cg) || use_solo_fused(stage) ? "k_convp_whole" : "k_convx_whole") :
				conv_path(eff_geom(stage)) == kPathGeneric ? "k_conv" :
				(conv_path(eff_geom(stage)) == kPathPair || conv_path(eff_geom(stage)) == kPathPair3 ||
					conv_path(eff_geom(stage)) == kPathPairP3 ? "k_convp" : "k_convx");
			break;
		case kFrac: *kernel = sp.whole ? "k_whole" : "k_poly"; break;
		case kHBUp: *kernel = group_len(stage) > 1 ? "k_hbcascade" : "k_hbup"; break;
		case kHBDown: *kernel = fuse_hbconv(stage) ? "k_convp_hb" : (group_len(stage) > 1 ? "k_hbdcascade" : "k_hbdown"); break;
		}
	}
	d.ms_sum = 0.0;
	d.launches = 0;
	d.t_in = d.t_out = 0;
	return true;
}

std::string Engine::stage_symbol(size_t stage) const
{
	return stage < dev_.size() ? dev_[stage].symbol : std::string();
}

void Engine::clear()
{
	// ring contents need no reset: positions restart at 0 and every position >= 0 is rewritten
	// before it is read again, positions < 0 read as zero by construction
	plan_.clear();
	// (park_cur too: the output-ring use of the buffers -- launch_stage once = 4, launch_fused oring -- knows park[0] only)
	for (StageDev& d : dev_)
	{
		d.park_base = d.park_end = 0;
		d.park_cur = 0;
	}
}

// ---- checkpoint -------------------------------------------------------------------------------
// blob: StateHeader, then per stage a StageState followed by the stage's input ring (nch x
// ring_size doubles) when that ring exists (the ring between two fused stages never does)
namespace {
struct StateHeader
{
	char magic[8];
	unsigned long long config;
	long long nstages, nch;
};
struct StageState
{
	long long m, done, rpos, ring_size, has_ring;
	double pos_frac, pos_shift;
	long long in_counter, in_pos_int;
	// parked outputs (Engine::stage_parks): doubles per channel (0: the stage does not park), the stream positions the
	// buffer holds; the buffer itself (nch x park_len doubles) follows the ring
	long long park_len, park_base, park_end;
};
}

unsigned long long Engine::config_hash() const
{
	// FNV-1a over everything that shapes the rings and the schedule
	std::string key = plan_.describe();
	key += "|maxin=" + std::to_string(plan_.max_in) + "|nch=" + std::to_string(nch_);
	for (const auto& kv : opt_)
		// (options that change neither the state nor a single bit of the stream stay out of it)
		if (kv.first != "timing" && kv.first != "walk" && kv.first != "walk_len")
			key += "|" + kv.first + "=" + std::to_string(kv.second);
	unsigned long long h = 1469598103934665603ull;
	for (unsigned char c : key)
	{
		h ^= c;
		h *= 1099511628211ull;
	}
	return h;
}

// does stage s ever own an input ring?  (the ring between the two stages of a fused pair / inside
// a fused run is never touched)
bool Engine::stage_owns_ring(size_t s) const
{
	size_t g = 0;
	while (g < plan_.stages.size())
	{
		if (g == s) return true;
		g += (size_t) group_len(g);
	}
	return false;
}

size_t Engine::state_size() const
{
	// the same before and after the first process() call: every ring a stage will ever own counts,
	// allocated yet or not (save_state allocates the missing ones, their content is all zero)
	size_t n = sizeof(StateHeader);
	for (size_t s = 0; s < dev_.size(); s++)
	{
		n += sizeof(StageState);
		if (stage_owns_ring(s)) n += (size_t) dev_[s].ring_size * (size_t) nch_ * sizeof(double);
		if (stage_parks(s)) n += (size_t) park_row_len(s) * (size_t) nch_ * sizeof(double);
	}
	return n;
}

size_t Engine::save_state(void* buf, size_t cap, void* stream)
{
	const size_t need = state_size();
	if (cap < need) throw std::runtime_error("state buffer too small");
	DevGuard guard(device_);
	for (size_t s = 0; s < dev_.size(); s++)
	{
		if (stage_owns_ring(s)) ensure_ring(s);
		if (stage_parks(s)) ensure_park(s);
	}
	dev_sync(stream);
	unsigned char* p = static_cast<unsigned char*>(buf);
	StateHeader h;
	std::memcpy(h.magic, "R8BHIPS2", 8);
	h.config = config_hash();
	h.nstages = (long long) dev_.size();
	h.nch = nch_;
	std::memcpy(p, &h, sizeof(h));
	p += sizeof(h);
	for (size_t s = 0; s < dev_.size(); s++)
	{
		const StagePlan& sp = plan_.stages[s];
		const StageDev& d = dev_[s];
		StageState st;
		st.m = sp.m; st.done = sp.done;
		st.rpos = sp.poly.rpos; st.pos_frac = sp.poly.pos_frac; st.pos_shift = sp.poly.pos_shift;
		st.in_counter = sp.poly.in_counter; st.in_pos_int = sp.poly.in_pos_int;
		st.ring_size = d.ring_size;
		st.has_ring = stage_owns_ring(s) ? 1 : 0;
		st.park_len = stage_parks(s) ? park_row_len(s) : 0;
		if (st.park_len != 0 && st.park_len != d.park_stride) throw std::logic_error("park buffer length");
		st.park_base = d.park_base; st.park_end = d.park_end;
		std::memcpy(p, &st, sizeof(st));
		p += sizeof(st);
		if (st.has_ring)
		{
			const size_t bytes = (size_t) d.ring_size * (size_t) nch_ * sizeof(double);
			dev_download(p, d.ring, bytes, stream);
			p += bytes;
		}
		if (st.park_len > 0)
		{
			const size_t bytes = (size_t) st.park_len * (size_t) nch_ * sizeof(double);
			if ((size_t) (p - static_cast<unsigned char*>(buf)) + bytes > need) throw std::logic_error("state size");
			dev_download(p, d.park[d.park_cur], bytes, stream);
			p += bytes;
		}
	}
	if ((size_t) (p - static_cast<unsigned char*>(buf)) != need) throw std::logic_error("state size");
	return need;
}

void Engine::load_state(const void* buf, size_t size, void* stream)
{
	const unsigned char* p = static_cast<const unsigned char*>(buf);
	const unsigned char* const end = p + size;
	StateHeader h;
	if (size < sizeof(h)) throw std::runtime_error("state blob truncated");
	std::memcpy(&h, p, sizeof(h));
	p += sizeof(h);
	if (std::memcmp(h.magic, "R8BHIPS2", 8) != 0) throw std::runtime_error("not a state blob");
	if (h.config != config_hash() || h.nstages != (long long) dev_.size() || h.nch != nch_)
		throw std::runtime_error("state blob was saved by a differently configured resampler");
	// pass 1: the whole blob against this object, nothing touched yet (a truncated or foreign blob
	// must not leave a half-restored stream behind)
	if (size != state_size()) throw std::runtime_error("state blob has the wrong size");
	std::vector<StageState> sts(dev_.size());
	std::vector<const unsigned char*> rings(dev_.size(), nullptr), parks(dev_.size(), nullptr);
	for (size_t s = 0; s < dev_.size(); s++)
	{
		const StageDev& d = dev_[s];
		StageState& st = sts[s];
		if ((size_t) (end - p) < sizeof(st)) throw std::runtime_error("state blob truncated");
		std::memcpy(&st, p, sizeof(st));
		p += sizeof(st);
		if (st.ring_size != d.ring_size) throw std::runtime_error("state blob ring size mismatch");
		if ((st.has_ring != 0) != stage_owns_ring(s))
			throw std::runtime_error("state blob ring layout mismatch");
		if (st.m < 0 || st.done < 0 || st.in_counter < 0 || st.in_pos_int < 0 ||
			st.in_counter > INT_MAX || st.in_pos_int > INT_MAX || !(st.pos_frac >= 0.0 && st.pos_frac < 1.0))
			throw std::runtime_error("state blob holds impossible counters");
		if (st.has_ring)
		{
			const size_t bytes = (size_t) d.ring_size * (size_t) nch_ * sizeof(double);
			if ((size_t) (end - p) < bytes) throw std::runtime_error("state blob truncated");
			rings[s] = p;
			p += bytes;
		}
		if (st.park_len != (stage_parks(s) ? park_row_len(s) : 0))
			throw std::runtime_error("state blob park layout mismatch");
		// (a stage that writes its last block ahead into a ring has counters but no buffer: what lies between them is
		// at most one block's outputs -- a larger park_end would make the next calls skip their blocks and hand out
		// whatever the ring holds --, and only a convolver ever has any)
		if (st.park_base < 0 || st.park_end < st.park_base || (st.park_len > 0 && st.park_end - st.park_base > st.park_len))
			throw std::runtime_error("state blob holds impossible counters");
		if (st.park_len == 0 && st.park_end - st.park_base >
			(plan_.stages[s].desc.kind == kConv && s + 1 < dev_.size() ? park_len_of(s, false) : 0))
			throw std::runtime_error("state blob holds impossible counters");
		if (st.park_len > 0)
		{
			const size_t bytes = (size_t) st.park_len * (size_t) nch_ * sizeof(double);
			if ((size_t) (end - p) < bytes) throw std::runtime_error("state blob truncated");
			parks[s] = p;
			p += bytes;
		}
	}
	if (p != end) throw std::runtime_error("state blob has trailing bytes");
	// pass 2: commit
	DevGuard guard(device_);
	dev_sync(stream);
	for (size_t s = 0; s < dev_.size(); s++)
	{
		StagePlan& sp = plan_.stages[s];
		const StageState& st = sts[s];
		sp.m = st.m; sp.done = st.done;
		sp.poly.rpos = st.rpos; sp.poly.pos_frac = st.pos_frac; sp.poly.pos_shift = st.pos_shift;
		sp.poly.in_counter = (int) st.in_counter; sp.poly.in_pos_int = (int) st.in_pos_int;
		if (rings[s] != nullptr)
		{
			ensure_ring(s);
			dev_upload(dev_[s].ring, rings[s], (size_t) dev_[s].ring_size * (size_t) nch_ * sizeof(double));
		}
		if (parks[s] != nullptr)
		{
			// (buffers of an earlier life under other options are replaced here -- while the old counters still say
			// whether they may be)
			dev_[s].park_base = dev_[s].park_end = 0;
			ensure_park(s);
		}
		dev_[s].park_base = st.park_base; dev_[s].park_end = st.park_end;
		if (parks[s] != nullptr)
		{
			dev_upload(dev_[s].park[dev_[s].park_cur], parks[s], (size_t) dev_[s].park_stride * (size_t) nch_ * sizeof(double));
		}
	}
}

void Engine::launch_stage(size_t s, long long m_prev, long long a, long long b,
	const PolyState& ps, const SrcView& src, const DstView& dst_in, void* stream)
{
	const StagePlan& sp = plan_.stages[s];
	const StageDev& d = dev_[s];
	(void) m_prev;
	DstView dst = dst_in;
	// emitted sample q is sample q + out_skip of the stage's stream function (fractional-latency chains
	// only): compute the shifted range, store it out_skip positions earlier
	a += sp.out_skip;
	b += sp.out_skip;
	dst.off -= sp.out_skip;
	switch (sp.desc.kind)
	{
	case kConv:
	{
		const ConvGeom g = eff_geom(s);
		ConvxLaunch X;
		ConvLaunch& L = X.c;
		fill_conv(s, L, src);
		if (hb_front_ >= 0)
		{
			// (launch_hbconv: `src` is the decimator's input stream; kernel mode 20 has no per-block spans, the front's
			// parameters lie over the end of that array)
			const StagePlan& hp = plan_.stages[(size_t) hb_front_];
			HbFront& F = X.hbf.p;
			F.n = hp.hb_n;
			F.np = hp.hb_n <= 4 ? 4 : (hp.hb_n <= 8 ? 8 : kHbfTapsMax);
			F.end = hp.m;
			for (int i = 0; i < kHbfTapsMax; i++) F.taps[i] = i < hp.hb_n ? hp.hb_taps[i] : 0.0;
			L.vec_ok = 0;
		}
		if (g.poly3)
		{
			// polyphase 3x form (ConvGeom::p3; r8b_convp.h mode 19): a block is a window of p3_n INPUT samples, its valid
			// outputs the 3 p3_m virtual samples from k in_len + blk_off - fl2 (a multiple of 3) on; rot carries the
			// components' reach into the past
			L.bl2 = g.bl2; L.in_len = g.in_len; L.n_in = L.n_out = g.n_in;
			L.blk_stride = g.in_len; L.blk_offset = g.blk_off;
			L.rot = g.p3_b;
			L.hp = d.hp3; L.ptw = d.ptw3;
		}
		const int path = conv_path(g);
		// Every block once (pair kernels): the block that holds the call's last output is computed whole -- what it
		// holds beyond b goes ahead into the next stage's ring (once = 3: nobody reads it before it is due) or, at the
		// end of the chain, into the park buffer (once = 2; ConvxLaunch::park_*) -- and the next call starts behind it
		// instead of computing that block again (one block in 13.4 for 44100 -> 88200 at BASELINE's call size, one in
		// 7.1 for 88200 -> 44100, one in 6.4 for 48000 -> 32000; cf. launch_fused)
		// One-channel fast path (r8b_convx.h: 16384-point blocks and what else the pair form does not cover): its store
		// clips at L.b, so the same is had without a kernel change -- in the middle of a chain by moving L.b to the
		// block's end (once = 3), at the end of the chain by letting the kernel write into an OUTPUT RING of the stage's
		// own (once = 4: the park buffer used as a ring of max_out_len + one block's outputs) and copying the call's
		// outputs from there to the caller's rows (k_tail: 2 x 8 bytes per output more, for one block in 5.2 less at
		// 48000 -> 32000 with a 0.5 % transition band)
		int once = (path == kPathPair || path == kPathPair3 || path == kPathPairP3) ? conv_once(s, dst) : 0;
		if ((path == kPathConvx || path == kPathConvx3) && opt_.at("park") && dst.fmt == kPcmF64)
			once = s + 1 == plan_.stages.size() ? (dst.mask == -1 && stage_parks(s) ? 4 : 0) : (dst.mask != -1 ? 3 : 0);
		StageDev& dd = dev_[s];
		// (blk_off: 0 but for the polyphase 3x form, whose blocks start on multiples of 3 -- <= 0, so the sum stays >= 0)
		auto blk_of = [&](long long q) { return ((long long) g.down * q + g.fl2 - g.blk_off) / g.in_len; };
		auto blk_end = [&](long long k) // the first output block k does not hold
		{
			const long long v = (k + 1) * (long long) g.in_len + g.blk_off - g.fl2;
			return v <= 0 ? 0LL : (v + g.down - 1) / g.down;
		};
		X.park_n = 0; X.park_out = 0; X.park_slices = 0; X.park_j0 = 0; X.park_stride = 0;
		X.walk = 0;
		X.quad = opt_.at("quad") != 0 ? 1 : 0;
		X.half = opt_.at("half") == 2 || (opt_.at("half") == 1 && half_worth(s)) ? 1 : 0;
		X.half_fused = 0;
		X.park_src = nullptr; X.park_dst = nullptr;
		X.park_blk = SpanInfo();
		long long ca = a; // the first output this call has to compute
		if (once == 2 || once == 4) ensure_park(s);
		// (once = 4: where the kernel writes, and what copies the call's outputs out of it afterwards)
		DstView rdst = dst;
		auto ring_to_rows = [&]()
		{
			TailLaunch T;
			T.src.ring = dd.park[0] + (long long) ch0_ * dd.park_stride;
			T.src.ring_stride = dd.park_stride; T.src.ring_mask = dd.park_stride - 1;
			T.src.cur = T.src.ring; T.src.cur_stride = 0; T.src.cur_base = LLONG_MAX;
			T.src.cur_fmt = kPcmF64;
			T.p0 = a; T.p1 = b;
			T.ring = dst.p + dst.off; T.ring_stride = dst.stride; T.ring_mask = -1;
			T.nch = nchw_;
			launch_tail(T, stream);
		};
		if (once == 4)
		{
			rdst.p = dd.park[0] + (long long) ch0_ * dd.park_stride;
			rdst.stride = dd.park_stride;
			rdst.mask = dd.park_stride - 1;
			rdst.off = 0;
		}
		if (once != 0 && dd.park_end > a)
		{
			ca = std::min(dd.park_end, b);
			if (once == 2)
			{
				if (dd.park_base > a) throw std::logic_error("parked outputs start behind the call's first output");
				X.park_src = dd.park[dd.park_cur] + (long long) ch0_ * dd.park_stride + (a - dd.park_base);
				X.park_stride = dd.park_stride;
				X.park_j0 = a;
				X.park_n = (int) (ca - a);
				if (ch0_ == 0) stat_["park_calls"]++;
			}
		}
		if (ca >= b)
		{
			// nothing left to compute: the outputs are in the next stage's ring already, or come out of the park buffer
			if (once == 2)
			{
				TailLaunch T;
				T.src.ring = X.park_src; T.src.ring_stride = 0; T.src.ring_mask = 0;
				T.src.cur = X.park_src; T.src.cur_stride = X.park_stride; T.src.cur_base = a;
				T.src.cur_fmt = kPcmF64;
				T.p0 = a; T.p1 = b;
				T.ring = dst.p + dst.off; T.ring_stride = dst.stride; T.ring_mask = -1;
				T.nch = nchw_;
				launch_tail(T, stream);
				if (ch0_ == 0) stat_["park_only_calls"]++;
			}
			if (once == 4)
			{
				ring_to_rows();
				if (ch0_ == 0) stat_["park_only_calls"]++;
			}
			break;
		}
		L.k0 = blk_of(ca);
		const long long k1 = blk_of(b - 1);
		L.nblk = (int) (k1 - L.k0 + 1);
		L.a = ca; L.b = b;
		L.dst = once == 4 ? rdst : dst;
		long long pend = b; // end of what the call's last block holds
		if (once != 0)
		{
			pend = blk_end(k1);
			if (pend < b) throw std::logic_error("block bookkeeping of the convolver");
			if (once == 3)
			{
				// (ahead into the ring: the ring was sized for it -- Engine::Engine)
				if (dst.mask == -1 || dst.mask + 1 < stage_history(s + 1) + plan_.stage_max_in[s + 1] + (pend - b))
					throw std::logic_error("ring too small for a block written ahead");
				L.b = pend;
			}
			if (once == 4)
			{
				if (pend - a > dd.park_stride) throw std::logic_error("output ring too small");
				L.b = pend;
			}
		}
		if (path != kPathGeneric)
		{
			X.in_step = X.out_step = 1; X.flen = 2; X.fl2w = X.fllw = 0; X.run_off = 0;
			X.ptab = nullptr; X.ctab = nullptr; X.nsets = 0;
			X.table = nullptr; X.wtab = nullptr; X.wa = X.wb = 0; X.wdst = L.dst;
			if (once == 2 && pend > b)
			{
				if (pend - b > dd.park_stride) throw std::logic_error("park buffer too small");
				X.park_out = 1;
				X.park_dst = dd.park[dd.park_cur ^ 1] + (long long) ch0_ * dd.park_stride;
				X.park_stride = dd.park_stride;
				X.park_blk.jlo = b;
				X.park_blk.jhi = pend;
			}
			if (path == kPathPairP3 && L.tail_ring != nullptr)
			{
				// (history for the next call, exactly: its first block is the one behind this call's last -- or the one that
				// holds output b --, whose window starts p3_b input samples before its first output's input position)
				const long long kn = once != 0 ? k1 + 1 : blk_of(b);
				const long long wstart = (kn * (long long) g.in_len + g.blk_off - g.fl2) / 3 - g.p3_b;
				const long long p0 = std::min(std::max(L.tail_p0, wstart - 8), L.tail_p1);
				L.tail_p0 = p0 < 0 ? 0 : (p0 & ~1LL);
			}
			if ((path == kPathPair || path == kPathPair3) && L.tail_ring != nullptr && g.up_pow2)
			{
				// (history for the next call, exactly -- cf. launch_fused: the next call's first block is the one that
				// holds output b -- the one behind this call's last when every block is computed once --, blocks sit at
				// multiples of in_len, a block's window is n_in input samples ending in_len / up behind its start)
				// (up is 1 or 2 on this path: convp_geometry_ok)
				if (g.up > 2) throw std::logic_error("pair convolver: up-sampling factor");
				const long long kn = once != 0 ? k1 + 1 : blk_of(b);
				const long long wstart = ((kn * g.in_len) >> (g.up > 1 ? 1 : 0)) - ((long long) g.n_in - g.in_len / g.up);
				const long long p0 = std::min(std::max(L.tail_p0, wstart - 8), L.tail_p1);
				L.tail_p0 = p0 < 0 ? 0 : (p0 & ~1LL);
			}
			if (ch0_ == 0) stat_["conv_blocks"] += L.nblk;
			const bool sp = convp_split_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2);
			const bool solo = convp_solo_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2, g.in_len) ||
				((!g.complex_h || g.down == 2) &&
					convp_solo_down_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2, g.in_len));
			// (long-block forms: + 4 with a complex kernel spectrum)
			const int cxl = g.complex_h ? 4 : 0;
			if (path == kPathPairP3) launch_convp(X, 19, stream);
			else if (path == kPathPair3) launch_convp(X, solo ? 11 + cxl : (sp ? 9 + cxl : (g.complex_h ? 7 : 3)), stream);
			else if (path == kPathConvx3) launch_convx(X, 3, stream);
			else if (path == kPathPair && hb_front_ >= 0) launch_convp(X, 20, stream);
			else if (path == kPathPair) launch_convp(X, solo ? 10 + cxl : (sp ? 8 + cxl : (g.complex_h ? 6 : 0)), stream);
			else launch_convx(X, 0, stream);
			if (L.tail_ring != nullptr) tail_done_ = true;
			if (once == 4) ring_to_rows();
			if (once != 0 && ch0_ + nchw_ >= nch_)
			{
				// (the counters once per call, after its last channel window)
				if (once == 2 && pend > b) dd.park_cur ^= 1;
				dd.park_base = b;
				dd.park_end = pend;
			}
			if (hb_front_ >= 0)
			{
				// (the next call's first block: the one behind this call's last when every block is computed once, else the
				// one that holds output b; its window starts n_in - in_len convolver inputs before the block)
				const long long kn = once != 0 ? k1 + 1 : blk_of(b);
				hb_next_raw_ = 2 * (kn * (long long) g.in_len - ((long long) g.n_in - g.in_len)) - 2 * kHbfTapsMax - 8;
			}
		}
		else
		{
			L.tail_ring = nullptr;
			if (generic_conv_big(g))
			{
				// (the reference's 32768-point block in front of a decimation in the spectrum: the launch's workgroups --
				// one per CU, 128 KB of LDS each -- walk the (block, channel) items; a workgroup's packed backward spectrum
				// passes through its own array in global memory)
				const long long items = (long long) L.nblk * L.nch;
				const int slots = (int) std::min<long long>(items, 256);
				ensure_work(s, slots, stream);
				L.work = dev_[s].work;
				L.work_slots = slots;
				L.threads = 512;
			}
			launch_conv(L, stream);
		}
		break;
	}
	case kFrac:
		if (sp.whole)
		{
			WholeLaunch L;
			L.in_step = sp.in_step; L.out_step = sp.out_step; L.flen = sp.flen;
			L.fl2 = sp.fl2; L.fll = sp.fll;
			L.pos0 = sp.pos0;
			L.table = d.table;
			// transposed table in output order (built for whole-step stages at construction): usable when
			// in_step is invertible mod out_step and the product phase * inverse stays in 32 bits
			L.wtab = nullptr; L.inv_in = 0;
			if (d.wtab != nullptr && sp.out_step < 46340)
			{
				long long inv = 0;
				for (long long k = 1; k < sp.out_step; k++) // (out_step is a few hundred at most; once per call)
					if (k * (sp.in_step % sp.out_step) % sp.out_step == 1) { inv = k; break; }
				if (sp.out_step == 1) inv = 0;
				if (inv != 0 || sp.out_step == 1) { L.wtab = d.wtab; L.inv_in = (int) inv; }
			}
			L.a = a; L.b = b;
			// as large as keeps the tile's input span within 32 KB of LDS (four to five workgroups per CU): a
			// thread's set-up -- a 64-bit division and its row fetch -- is paid once per tile (44100 -> 96000,
			// 1024 channels: tile 1024 0.237 ms, 2048 0.191, 4096 0.156)
			L.tile = opt_.at("whole_tile");
			L.span_max = (int) ((long long) L.tile * sp.in_step / sp.out_step) + sp.flen + 4 + 32;
			while ((L.span_max > 4096 && L.tile > 1024) || (L.span_max > 12288 && L.tile > 64))
			{
				L.tile /= 2;
				L.span_max = (int) ((long long) L.tile * sp.in_step / sp.out_step) + sp.flen + 4 + 32;
			}
			L.nch = nchw_;
			L.src = src; L.dst = dst;
			launch_whole(L, stream);
		}
		else
		{
			PolyLaunch L;
			L.flen = sp.flen; L.fl2 = sp.fl2; L.fll = sp.fll; L.fracs = sp.bank->fracs;
			L.table = d.table;
			L.ssr = sp.ssr; L.dsr = sp.dsr;
			L.rpos0 = ps.rpos; L.fpos0 = ps.pos_frac;
			L.counter0 = ps.in_counter; L.pos_int0 = ps.in_pos_int; L.shift = ps.pos_shift;
			L.a = a; L.b = b; L.nch = nchw_;
			// tile of 64 outputs spans at most 64*Src/Dst + taps input samples (+ slack for the
			// counter's rounding)
			L.span_max = opt_.at("poly_tiled") ?
				(int) std::ceil(64.0 * sp.ssr / sp.dsr) + sp.flen + 4 + 8 : 0; // (+ kPolyPad zeros)
			L.pitch = poly_row_pitch(L.span_max, sp.ssr / sp.dsr);
			L.front = L.span_max > 0 && L.span_max - 8 <= 16 * 12; // (kPolyPad, kPolyNV)
			L.src = src; L.dst = dst;
			launch_poly(L, stream);
		}
		break;
	case kHBUp:
	case kHBDown:
	{
		HBLaunch L;
		L.ntaps = sp.hb_n;
		if (sp.hb_n > 16) throw std::runtime_error("half-band filter too long");
		for (int i = 0; i < 16; i++) L.taps[i] = i < sp.hb_n ? sp.hb_taps[i] : 0.0;
		L.a = a; L.b = b;
		L.tile = opt_.at("hb_tile");
		L.nch = nchw_;
		L.src = src; L.dst = dst;
		take_carried_tail(L.tail, &L.carry_tail);
		if (sp.desc.kind == kHBUp) launch_hbup(L, stream);
		else launch_hbdown(L, stream);
		break;
	}
	}
}

int Engine::process_planar(const void* d_in, int in_fmt, long long in_stride, int l, void* d_out,
	int out_fmt, long long out_stride, void* stream)
{
	// The first stage decodes the caller's samples as it loads them, the last one encodes as it stores (src_load /
	// dst_store: no staging copy, 2-4 bytes per sample at the HBM edge) -- WHERE that stage's kernel exists for PCM
	// views: the streaming kernels and the generic convolver (pcm_fused_in / pcm_fused_out).  The compile-time-sized
	// convolvers are built for fp64 views only; a PCM side in front of / behind one of them has to arrive as fp64 rows
	// (r8b_capi.cpp batch_process_pcm decodes / encodes through the object's staging rows, one extra pass over that
	// side).  Refused here, for every caller, instead of failing inside a launcher.
	if (in_fmt != kPcmF64 && !pcm_fused_in())
		throw std::runtime_error("planar PCM input in front of a compile-time-sized convolver: decode it into fp64 rows "
			"first (r8b_batch_process_pcm does)");
	if (out_fmt != kPcmF64 && !pcm_fused_out())
		throw std::runtime_error("planar PCM output behind a compile-time-sized convolver: take fp64 rows and encode "
			"them (r8b_batch_process_pcm does)");
	struct Reset
	{
		Engine& e;
		~Reset() { e.io_in_fmt_ = e.io_out_fmt_ = kPcmF64; }
	} reset{*this};
	io_in_fmt_ = in_fmt;
	io_out_fmt_ = out_fmt;
	return process(static_cast<const double*>(d_in), in_stride, l, static_cast<double*>(d_out),
		out_stride, stream);
}

int Engine::process(const double* d_in, long long in_stride, int l, double* d_out,
	long long out_stride, void* stream)
{
	if (l < 0 || l > plan_.max_in) throw std::runtime_error("input length exceeds MaxInLen");
	if (l == 0) return 0;
	// raw device pointers from a foreign host: refuse what would read or write outside the rows
	if (d_in == nullptr || d_out == nullptr) throw std::runtime_error("null device pointer");
	if (nch_ > 1 && (in_stride < l || in_stride < 0))
		throw std::runtime_error("in_stride is smaller than the number of input samples per row");
	{
		// what this call will return (the plan is advanced only by the stage loop below)
		ChainPlan probe = plan_;
		int n = l;
		for (StagePlan& sp : probe.stages)
		{
			long long a, b;
			sp.step(n, &a, &b, nullptr);
			n = (int) (b - a);
		}
		if (plan_.stages.empty()) n = l;
		if (nch_ > 1 && out_stride < n)
			throw std::runtime_error("out_stride is smaller than the number of output samples of "
				"this call (" + std::to_string(n) + "): rows would overlap");
	}
	if ((io_in_fmt_ == kPcmF64 && ((size_t) d_in & 7) != 0) || (io_out_fmt_ == kPcmF64 && ((size_t) d_out & 7) != 0))
		throw std::runtime_error("fp64 buffers must be 8-byte aligned");
	DevGuard guard(device_);
	const size_t ns = plan_.stages.size();
	if (ns == 0)
	{
		// Src == Dst: the reference hands the input back (reference CDSPResampler.h:534-535);
		// the batch entry copies it into the caller's output buffer
		TailLaunch T;
		T.src.ring = d_in; T.src.ring_stride = 0; T.src.ring_mask = 0;
		T.src.cur = d_in; T.src.cur_stride = in_stride; T.src.cur_base = 0;
		T.src.cur_fmt = kPcmF64;
		if (io_in_fmt_ != kPcmF64 || io_out_fmt_ != kPcmF64)
			throw std::logic_error("pass-through of PCM buffers goes through the staging rows");
		T.p0 = 0; T.p1 = l;
		T.ring = d_out; T.ring_stride = out_stride; T.ring_mask = -1;
		T.nch = nch_;
		launch_tail(T, stream);
		return l;
	}
	// Pass 1 (host only): advance the plan, stage group by stage group; what each launch has to produce.
	struct Rec
	{
		size_t s;
		int glen;
		long long m_prev, a, b, wa, wb;
		PolyState ps;
		bool fused, work;
	};
	std::vector<Rec> recs;
	int n = l;
	for (size_t s = 0; s < ns; s++)
	{
		StagePlan& sp = plan_.stages[s];
		Rec r;
		r.s = s;
		r.m_prev = sp.m;
		sp.step(n, &r.a, &r.b, &r.ps);
		// stages [s, s+glen) are executed by one launch: convolver + whole-step interpolator, or a
		// run of half-band up-samplers
		r.glen = group_len(s);
		r.fused = r.glen > 1;
		r.wa = r.a;
		r.wb = r.b;
		for (int g = 1; g < r.glen; g++)
		{
			long long ga, gb;
			plan_.stages[s + g].step((int) (r.wb - r.wa), &ga, &gb, nullptr);
			r.wa = ga;
			r.wb = gb;
		}
		r.work = r.fused ? r.wb > r.wa : r.b > r.a;
		n = (int) (r.fused ? r.wb - r.wa : r.b - r.a);
		recs.push_back(r);
		s += r.glen - 1;
	}
	// Pass 2: the launches, for the channel window [ch0_, ch0_ + nchw_).
	auto launch_rec = [&](const Rec& r)
	{
		const size_t s = r.s;
		const StagePlan& sp = plan_.stages[s];
		const size_t last = s + r.glen - 1; // stage whose output this launch produces
		SrcView src;
		src.ring = dev_[s].ring + (long long) ch0_ * dev_[s].ring_size;
		src.ring_stride = dev_[s].ring_size;
		src.ring_mask = dev_[s].ring_size - 1;
		if (s == 0)
		{
			src.cur = d_in + (long long) ch0_ * in_stride; // (windows are used with fp64 buffers only)
			src.cur_stride = in_stride;
			src.cur_base = r.m_prev;
			src.cur_fmt = io_in_fmt_;
		}
		else
		{
			src.cur = nullptr;
			src.cur_stride = 0;
			src.cur_base = LLONG_MAX;
			src.cur_fmt = kPcmF64;
		}
		DstView dst;
		if (last + 1 == ns)
		{
			dst.p = d_out + (long long) ch0_ * out_stride;
			dst.stride = out_stride;
			dst.mask = -1;
			dst.off = -(r.fused ? r.wa : r.a);
			dst.fmt = io_out_fmt_;
		}
		else
		{
			dst.p = dev_[last + 1].ring + (long long) ch0_ * dev_[last + 1].ring_size;
			dst.stride = dev_[last + 1].ring_size;
			dst.mask = dev_[last + 1].ring_size - 1;
			dst.off = 0;
			dst.fmt = kPcmF64;
		}
		if (s == 0)
		{
			tail_done_ = false;
			// History for the next call: the last history() samples of the stream go into the OTHER ring (this call's
			// kernels may still be reading the current one).  The fast convolver does the copy itself (tail_done_);
			// otherwise it is CARRIED by the call's next half-band launch -- stage 0's own or a later one's: extra
			// workgroups of that grid, nobody reads the other ring in this call -- and only when the call has none by
			// the copy kernel (k_tail, below and behind the stage loop).  (fp64 caller rows, the whole batch in one
			// launch: a later stage's kernel is the fp64 build, a channel window has its own tail.)
			carry_ = false;
			carry_tail_.src = src;
			carry_tail_.p1 = sp.m;
			carry_tail_.p0 = sp.m - stage_history(0);
			if (carry_tail_.p0 < 0) carry_tail_.p0 = 0;
			carry_tail_.ring = dev_[0].ring_alt + (long long) ch0_ * dev_[0].ring_size;
			carry_tail_.ring_stride = dev_[0].ring_size;
			carry_tail_.ring_mask = dev_[0].ring_size - 1;
			carry_tail_.nch = nchw_;
			carry_ = opt_.at("fold_tail") != 0 && src.cur_fmt == kPcmF64 && nchw_ == nch_;
		}
		if (r.work)
		{
			const bool timing = opt_.at("timing") != 0;
			void *e0 = nullptr, *e1 = nullptr;
			if (timing)
			{
				e0 = get_event(dev_[s]);
				e1 = get_event(dev_[s]);
				dev_event_record(e0, stream);
			}
			if (r.fused && sp.desc.kind == kConv) launch_fused(s, r.wa, r.wb, src, dst, stream);
			else if (r.fused && sp.desc.kind == kHBDown && fuse_hbconv(s)) launch_hbconv(s, r.wa, r.wb, src, dst, stream);
			else if (r.fused && sp.desc.kind == kHBDown) launch_dcascade(s, r.glen, r.wa, r.wb, src, dst, stream);
			else if (r.fused) launch_cascade(s, r.glen, r.wa, r.wb, src, dst, stream);
			else launch_stage(s, r.m_prev, r.a, r.b, r.ps, src, dst, stream);
			if (timing)
			{
				dev_event_record(e1, stream);
				dev_[s].pending.emplace_back(e0, e1);
				// (what the launcher really started for this stage: rocprofv3's name for it)
				if (const char* sym = launch_symbol_last()) dev_[s].symbol = sym;
				// (one (in, out) count per call, however many channel windows it is launched in)
				if (ch0_ == 0)
				{
					dev_[s].t_in += (long long) (sp.m - r.m_prev);
					dev_[s].t_out += r.fused ? r.wb - r.wa : r.b - r.a;
				}
			}
		}
		if (s == 0 && tail_done_) carry_ = false;
		if (s == 0 && !tail_done_ && !carry_)
		{
			launch_tail(carry_tail_, stream);
			if (ch0_ == 0) stat_["tail_launches"]++;
		}
	};
	for (const Rec& r : recs)
	{
		ensure_ring(r.s);
		if (r.s + r.glen < ns) ensure_ring(r.s + r.glen);
	}
	ch0_ = 0;
	nchw_ = nch_;
	for (size_t i = 0; i < recs.size(); i++)
	{
		// A convolver followed by the polynomial interpolator (non-whole-step ratios, 44100 -> 44101): two launches
		// with the convolver's whole 2x stream between them -- 268 MB for BASELINE's batch, written to HBM and read
		// back.  Walked in channel groups whose stream (<= ~96 MB) stays in the 256 MB Infinity Cache between the
		// two launches, the interpolator reads it from there (reference CDSPFracInterpolator.h:1069-1179 behind
		// CDSPBlockConvolver.h:252-354).  Channels are independent, so results do not change.
		const Rec& r = recs[i];
		int groups = 1;
		if (i + 1 < recs.size() && opt_.at("poly_groups") && !r.fused && !recs[i + 1].fused && r.work && recs[i + 1].work &&
			plan_.stages[r.s].desc.kind == kConv && plan_.stages[recs[i + 1].s].desc.kind == kFrac &&
			!plan_.stages[recs[i + 1].s].whole && io_in_fmt_ == kPcmF64 && io_out_fmt_ == kPcmF64)
		{
			const double between = 8.0 * (double) nch_ * (double) (r.b - r.a);
			const double cap = opt_.at("poly_groups") > 1 ? 1024.0 * opt_.at("poly_groups") : 96.0 * 1048576.0;
			groups = (int) std::ceil(between / cap);
			// (whole channel pairs per group, at least 256 pairs each: smaller launches do not fill the chip)
			// (an explicit cap -- tests -- may cut down to single pairs)
			const int min_per = opt_.at("poly_groups") > 1 ? 2 : 512;
			while (groups > 1 && (nch_ / groups) < min_per) groups--;
		}
		if (groups <= 1)
		{
			launch_rec(r);
			continue;
		}
		const int per = ((nch_ + groups - 1) / groups + 1) & ~1;
		for (int c0 = 0; c0 < nch_; c0 += per)
		{
			ch0_ = c0;
			nchw_ = std::min(per, nch_ - c0);
			launch_rec(r);
			launch_rec(recs[i + 1]);
		}
		ch0_ = 0;
		nchw_ = nch_;
		i++;
	}
	if (carry_)
	{
		// (no launch of this call could carry the history copy)
		carry_ = false;
		launch_tail(carry_tail_, stream);
		stat_["tail_launches"]++;
	}
	if (ns > 0) std::swap(dev_[0].ring, dev_[0].ring_alt);
	return n;
}

// Half-band decimator s in front of convolver s + 1 as one launch (r8b_convp.h mode 20): linear-phase chains, the
// 4096 -> 2048-point decimating geometry of the pair kernel (176400 -> 44100, 192000 -> 48000 ... at the 24-bit preset).
bool Engine::fuse_hbconv(size_t s) const
{
	return opt_.at("fuse_hbconv") && opt_.at("fast_conv") && opt_.at("pair_conv") && hbconv_possible(s) &&
		conv_path(plan_.stages[s + 1].cg) == kPathPair && use_pair(plan_.stages[s + 1].cg);
}

// ... whatever the options say (the rings are sized once, for either form)
bool Engine::hbconv_possible(size_t s) const
{
	if (s + 1 >= plan_.stages.size()) return false;
	const StagePlan& h = plan_.stages[s];
	const StagePlan& c = plan_.stages[s + 1];
	if (h.desc.kind != kHBDown || c.desc.kind != kConv || latency_chain()) return false;
	if (h.out_skip != 0 || h.hb_n > kHbfTapsMax || h.hb_n < 1) return false;
	const ConvGeom& g = c.cg;
	if (g.poly3 || g.complex_h || !g.up_pow2 || g.up != 1 || !g.down_pow2 || g.down != 2) return false;
	return g.n_in == 2 * kHbfRound && g.n_out == kHbfRound && (g.in_len & 1) == 0;
}

void Engine::launch_hbconv(size_t s, long long wa, long long wb, const SrcView& src, const DstView& dst, void* stream)
{
	struct Reset
	{
		long long& v;
		~Reset() { v = -1; }
	} reset{hb_front_};
	hb_front_ = (long long) s;
	hb_next_raw_ = LLONG_MIN;
	launch_stage(s + 1, 0, wa, wb, PolyState(), src, dst, stream);
	// History for the next call (stage 0: the caller's buffer is gone then): the raw stream from where the next call's
	// first block starts reading -- not the whole stage_history() the pending copy was set up with
	if (s == 0 && hb_next_raw_ != LLONG_MIN && (carry_ || !tail_done_))
	{
		long long p0 = std::max(carry_tail_.p0, hb_next_raw_);
		p0 = std::min(p0, carry_tail_.p1);
		carry_tail_.p0 = p0 < 0 ? 0 : p0;
	}
}

int Engine::group_len(size_t s) const
{
	// (chains with a fractional latency: the convolver + interpolator pair -- fuse_latency_ok -- and the half-band runs,
	// whose cascade kernels know the stages' skipped outputs)
	const bool lat = latency_chain();
	if (lat && !opt_.at("fuse_latency")) return 1;
	if (fuse_with_next(s)) return 2;
	if (fuse_hbconv(s)) return 2;
	const StageKind kind = plan_.stages[s].desc.kind;
	if (lat && kind != kHBUp && kind != kHBDown) return 1;
	// Runs of decimators: one kernel saves two launches and the intermediate streams, but pays
	// ~25 % of halo recomputation and holds 25 KB of LDS per workgroup; measured on sacd.cpp's
	// 2822400 -> 176400 it wins on small batches (64 ch x 65536: 0.032 vs 0.037 ms) and loses on
	// large ones (256 ch: 0.111 vs 0.099 ms).  The choice must not change between calls (the
	// unfused stages keep their history in rings the fused kernel never writes): it is made from
	// the object's constants.
	const bool down_ok = opt_.at("fuse_hbd") == 1 || (opt_.at("fuse_hbd") == 2 &&
		(long long) nch_ * plan_.stage_max_in[s] < (8LL << 20));
	if (opt_.at("fuse_hb") && (kind == kHBUp || (kind == kHBDown && down_ok)))
	{
		int n = 1;
		// (a decimator that goes into the convolver behind it -- fuse_hbconv -- is not part of a run)
		while (s + n < plan_.stages.size() && plan_.stages[s + n].desc.kind == kind &&
			n < kMaxCascade && !fuse_hbconv(s + n)) n++;
		return n;
	}
	return 1;
}

// input samples a stage must keep from earlier calls.  A run of half-band decimators executed as
// one kernel looks back over the accumulated filter spans of the whole run.
long long Engine::hbconv_history(size_t s) const
{
	return 2LL * plan_.stages[s + 1].history() + 4 * kHbfTapsMax + 64;
}

long long Engine::stage_history(size_t s) const
{
	const StagePlan& sp = plan_.stages[s];
	if (sp.desc.kind != kHBDown) return sp.history();
	// (decimator + convolver as one launch: the convolver's history, in raw samples, + the decimator's own reach with its
	// taps rounded up + the samples the decimator has taken without an output yet)
	if (fuse_hbconv(s)) return hbconv_history(s);
	long long span = 0;
	int g = 0;
	while (s + g < plan_.stages.size() && plan_.stages[s + g].desc.kind == kHBDown && g < kMaxCascade)
	{
		const int T = plan_.stages[s + g].hb_n <= 4 ? 4 : (plan_.stages[s + g].hb_n <= 8 ? 8 : 14);
		span += (long long) (2 * T - 1) << g;
		g++;
	}
	return std::max<long long>(sp.history(), 2 * span + (2LL << g) + 64);
}

void Engine::launch_dcascade(size_t s, int glen, long long fa, long long fb, const SrcView& src,
	const DstView& dst_in, void* stream)
{
	HBCascadeLaunch L;
	L.nst = glen;
	L.has_skip = 0;
	for (int g = 0; g < kMaxCascade; g++)
	{
		L.ntaps[g] = 0;
		L.skip[g] = 0;
		for (int k = 0; k < 14; k++) L.taps[g][k] = 0.0;
	}
	// (chains with a fractional latency: the stages' skipped outputs, as in launch_cascade)
	for (int g = 0; g + 1 < glen; g++)
	{
		L.skip[g] = (int) plan_.stages[s + g].out_skip;
		if (L.skip[g] != 0) L.has_skip = 1;
	}
	const long long sk_last = plan_.stages[s + glen - 1].out_skip;
	fa += sk_last;
	fb += sk_last;
	DstView dst = dst_in;
	dst.off -= sk_last;
	for (int g = 0; g < glen; g++)
	{
		const StagePlan& sp = plan_.stages[s + g];
		if (sp.hb_n > 14) throw std::runtime_error("half-band filter too long");
		L.ntaps[g] = sp.hb_n <= 4 ? 4 : (sp.hb_n <= 8 ? 8 : 14);
		for (int k = 0; k < sp.hb_n; k++) L.taps[g][k] = sp.hb_taps[k];
	}
	L.a = fa; L.b = fb;
	// last-stage outputs per workgroup: about 4096 first-stage input samples
	int tile = std::max(32, opt_.at("hbd_span") >> glen);
	L.tile = tile;
	// LDS: stage inputs alternate between two buffers; size each for a full tile
	long long lo = 0, hi = tile, even = 0, odd = 0;
	for (int g = glen - 1; g >= 0; g--)
	{
		const int T = L.ntaps[g];
		const long long ilo = 2 * lo - (2 * T - 1), ihi = 2 * (hi - 1) + (2 * T - 1) + 1;
		lo = ilo;
		hi = ihi;
		long long& m = (g & 1) ? odd : even;
		m = std::max(m, hi - lo);
	}
	L.buf = (int) even + 8;
	L.buf2 = (int) odd + 8;
	L.pair_ok = 0;
	L.in_end = plan_.stages[s].m;
	L.nch = nchw_;
	L.src = src; L.dst = dst;
	take_carried_tail(L.tail, &L.carry_tail);
	launch_hbdcascade(L, stream);
}

void Engine::launch_cascade(size_t s, int glen, long long fa, long long fb, const SrcView& src,
	const DstView& dst_in, void* stream)
{
	HBCascadeLaunch L;
	L.nst = glen;
	for (int g = 0; g < kMaxCascade; g++)
	{
		L.ntaps[g] = 0;
		L.skip[g] = 0;
		for (int k = 0; k < 14; k++) L.taps[g][k] = 0.0;
	}
	// chains with a fractional latency: a stage emits its stream from output out_skip on.  Between the stages of the run
	// that is HBCascadeLaunch::skip; for the last one, as in launch_stage: the shifted range, stored out_skip earlier
	L.has_skip = 0;
	for (int g = 0; g + 1 < glen; g++)
	{
		L.skip[g] = (int) plan_.stages[s + g].out_skip;
		if (L.skip[g] != 0) L.has_skip = 1;
	}
	const long long sk_last = plan_.stages[s + glen - 1].out_skip;
	fa += sk_last;
	fb += sk_last;
	DstView dst = dst_in;
	dst.off -= sk_last;
	for (int g = 0; g < glen; g++)
	{
		const StagePlan& sp = plan_.stages[s + g];
		if (sp.hb_n > 14) throw std::runtime_error("half-band filter too long");
		// rounded up to the unrolled widths of the kernel (extra taps are zero)
		L.ntaps[g] = sp.hb_n <= 4 ? 4 : (sp.hb_n <= 8 ? 8 : 14);
		for (int k = 0; k < sp.hb_n; k++) L.taps[g][k] = sp.hb_taps[k];
	}
	hbc_fill_ranges(L);
	L.a = fa; L.b = fb;
	// last-stage outputs per workgroup: a multiple of 2^glen.  A tile costs ~1.7 us of a CU
	// whatever its size (six dependent phases), so large batches take 8192 (51 KB LDS, 3
	// workgroups per CU: 0.19 vs 0.22 ms on cfg5 x 1024 channels) and small ones 4096, which
	// keeps every CU busy
	int want = opt_.at("hbc_tile");
	if (want == 0) want = (fb - fa + 8191) / 8192 * (long long) nch_ >= 256 * 6 ? 8192 : 4096;
	int tile = 1 << glen;
	while (tile < want) tile <<= 1;
	L.tile = tile;
	L.buf = tile / 2 + 96;  // largest intermediate stream of a tile (input of the last stage)
	L.buf2 = tile / 4 + 96; // the one before it (the buffers alternate)
	L.nch = nchw_;
	L.src = src; L.dst = dst;
	L.in_end = plan_.stages[s].m;
	// 16-byte stores of output pairs: output q is even for the first of an (even, odd) pair, whose element
	// index is even when the offset is (1); with an odd offset the aligned pairs are (odd, next even) (2)
	L.pair_ok = dst.fmt == kPcmF64 && ((size_t) dst.p & 15) == 0 && (dst.stride & 1) == 0 ?
		((dst.off & 1) == 0 ? 1 : 2) : 0;
	take_carried_tail(L.tail, &L.carry_tail);
	launch_hbcascade(L, stream);
}

// which kernel family runs a (not fused) convolver stage
ConvGeom Engine::eff_geom(size_t s) const
{
	ConvGeom g = plan_.stages[s].cg;
	if (g.p3 && opt_.at("up3_poly") && opt_.at("pair_conv") && opt_.at("fast_conv"))
	{
		g.poly3 = true;
		g.in_len = 3 * g.p3_m;
		g.bl2 = 3 * g.p3_n;
		g.n_in = g.n_out = g.p3_n;
		g.blk_off = g.p3_off;
	}
	return g;
}

int Engine::conv_path(const ConvGeom& g) const
{
	if (g.poly3) return kPathPairP3;
	if (!(opt_.at("fast_conv") || !generic_conv_fits(g))) return kPathGeneric;
	// (8192 -> 16384-point blocks: the split 2x up-sampling form of the pair kernel, two channels per workgroup, instead
	// of the one-channel kernel)
	// (with a complex kernel spectrum -- modes 12 ... 15 -- these forms are the only path such blocks have when the generic
	// kernel's arrays do not fit: the options do not switch them off then)
	const bool cx_only = g.complex_h && !generic_conv_fits(g);
	if (((opt_.at("pair_conv") && opt_.at("pair_split")) || cx_only) &&
		convp_split_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2))
		return g.down == 3 ? kPathPair3 : kPathPair;
	// (16384-point blocks 1:1: the one-channel form of the pair kernel instead of the one-channel kernel)
	if (((opt_.at("pair_conv") && opt_.at("pair_solo")) || cx_only) && (!g.complex_h || g.n_in == g.n_out) &&
		convp_solo_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2, g.in_len))
		return (!g.up_pow2 && g.up == 3) || (!g.down_pow2 && g.down == 3) ? kPathPair3 : kPathPair;
	if (((opt_.at("pair_conv") && opt_.at("pair_solo")) || cx_only) && (!g.complex_h || g.down == 2) &&
		convp_solo_down_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2, g.in_len))
		return !g.up_pow2 && g.up == 3 ? kPathPair3 : kPathPair;
	if (opt_.at("pair_conv") && convp_mode3_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2))
		return kPathPair3;
	if (g.complex_h) return use_pair(g) ? kPathPair : kPathGeneric; // (complex spectrum: pair kernel or generic)
	if (convx_mode3_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2)) return kPathConvx3;
	if (use_pair(g)) return kPathPair;
	if (convx_geometry_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2)) return kPathConvx;
	return kPathGeneric;
}

// some compile-time-sized kernel (r8b_convx.h or r8b_convp.h) is instantiated for the geometry
bool Engine::fast_geometry(const ConvGeom& g) const
{
	return convx_geometry_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2) || use_pair(g);
}

bool Engine::use_pair_fused(const ConvGeom& g) const
{
	return opt_.at("pair_conv") && convp_fused_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2);
}

bool Engine::use_pair(const ConvGeom& g) const
{
	return opt_.at("pair_conv") &&
		convp_geometry_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2);
}

bool Engine::latency_chain() const
{
	for (const StagePlan& sp : plan_.stages)
		if (sp.out_skip != 0 || sp.pos0 != 0 || sp.frac0 != 0.0 || (sp.desc.kind == kConv && sp.cg.complex_h))
			return true;
	return false;
}

// PCM at the edges (Engine::process_planar): a streaming kernel or the generic convolver decodes / encodes planar PCM
// caller buffers in place; the compile-time-sized convolvers (r8b_convx.h, r8b_convp.h) exist for fp64 views only --
// r8b_kernels.hip, top -- and have the samples brought to them through the staging rows (r8b_capi.cpp).
bool Engine::pcm_fused_in() const
{
	if (plan_.stages.empty()) return false;
	const StagePlan& sp = plan_.stages[0];
	if (fuse_hbconv(0)) return false; // (the first launch is a compile-time-sized convolver)
	return !(sp.desc.kind == kConv && conv_path(eff_geom(0)) != kPathGeneric);
}

bool Engine::pcm_fused_out() const
{
	const size_t ns = plan_.stages.size();
	if (ns == 0) return false;
	if (ns >= 2 && fuse_with_next(ns - 2)) return false; // (convolver + whole-step interpolator as one fast-path kernel)
	const StagePlan& sp = plan_.stages[ns - 1];
	return !(sp.desc.kind == kConv && conv_path(eff_geom(ns - 1)) != kPathGeneric);
}

// The interpolator of a chain with a fractional latency emits sample j as sample q = j + out_skip of its stream, whose
// position is q In + pos0 (reference CDSPFracInterpolator.h:721-752, 991-1060); its input sample i is the convolver's
// output t = i + out_skip of THAT stage.  With j0 In = pos0 (mod Out) -- In and Out are coprime -- and m = (j0 In - pos0)
// / Out, floor((q In + pos0) / Out) = floor((q + j0) In / Out) - m and the phases agree: the stream is the canonical one
// (position J In, start phase 0) renumbered, J = j + out_skip_w + j0, read from convolver outputs floor(J In / Out) + d,
// d = out_skip_c - m.
Engine::FusedShift Engine::fused_shift(size_t s) const
{
	const StagePlan& c = plan_.stages[s];
	const StagePlan& w = plan_.stages[s + 1];
	FusedShift f;
	f.js = 0; f.d = 0; f.t_zero = 0;
	if (c.out_skip == 0 && w.out_skip == 0 && w.pos0 == 0) return f;
	const long long In = w.in_step, Out = w.out_step;
	long long j0 = 0;
	while (j0 < Out && (j0 * In) % Out != (long long) w.pos0 % Out) j0++;
	if (j0 >= Out) throw std::logic_error("fused_shift: the interpolator's start phase has no canonical output");
	const long long m = (j0 * In - w.pos0) / Out;
	f.js = w.out_skip + j0;
	f.d = c.out_skip - m;
	f.t_zero = (int) c.out_skip;
	return f;
}

// ... which needs the pair kernel with two phases per thread (modes 4 / 5 / 16 / 17 carry the shifts) and nothing else
// in the pair of stages that the fused launch does not model
bool Engine::fuse_latency_ok(size_t s) const
{
	if (!opt_.at("fuse_latency") || !opt_.at("pair_two") || s + 1 >= plan_.stages.size()) return false;
	const StagePlan& c = plan_.stages[s];
	const StagePlan& w = plan_.stages[s + 1];
	if (c.desc.kind != kConv || w.desc.kind != kFrac || !w.whole || w.frac0 != 0.0) return false;
	if (!use_pair_fused(c.cg) || c.out_skip < 0 || w.out_skip < 0 || w.pos0 < 0 || w.pos0 >= w.out_step) return false;
	if (!two_phase_possible(w, nullptr)) return false;
	// (use_pair_two's test of the run's place in the array, before the lane tables exist)
	const int off = (w.in_step + 16 + 15) / 16 * 16;
	return off + c.cg.in_len + w.in_step + 32 + 16 <= c.cg.n_out;
}

// A 1:1 convolver on 16384-point blocks in front of a whole-step interpolator: the pair kernel's one-channel form with
// the interpolator fused in (r8b_convp.h mode 18: real kernel spectrum, plain load)
bool Engine::use_solo_fused(size_t s) const
{
	if (!opt_.at("solo_fuse") || !opt_.at("pair_solo") || !opt_.at("pair_conv") || s + 1 >= plan_.stages.size()) return false;
	const StagePlan& c = plan_.stages[s];
	const StagePlan& w = plan_.stages[s + 1];
	if (c.desc.kind != kConv || w.desc.kind != kFrac || !w.whole) return false;
	const ConvGeom& g = c.cg;
	return !g.complex_h && g.up_pow2 && g.up == 1 && g.down == 1 && conv_path(g) == kPathPair &&
		convp_solo_ok(g.n_in, g.n_out, g.up, g.down, g.up_pow2, g.down_pow2, g.in_len);
}

bool Engine::fuse_with_next(size_t s) const
{
	if (latency_chain() && !fuse_latency_ok(s)) return false;
	if (!opt_.at("fuse") || !opt_.at("fast_conv") || s + 1 >= plan_.stages.size()) return false;
	const StagePlan& c = plan_.stages[s];
	const StagePlan& w = plan_.stages[s + 1];
	if (c.desc.kind != kConv || w.desc.kind != kFrac || !w.whole || c.cg.down != 1) return false;
	// (8192 -> 16384-point blocks: the pair kernel's split form + the unfused interpolator beat the fused one-channel kernel;
	// 16384-point 1:1 blocks: the pair kernel's one-channel form with the interpolator fused in -- use_solo_fused)
	if (use_solo_fused(s)) return w.flen <= 32 && c.cg.in_len >= 4 * w.flen && c.cg.in_len + 64 <= c.cg.n_out;
	if (conv_path(c.cg) == kPathPair && !use_pair(c.cg)) return false;
	if (!convx_geometry_ok(c.cg.n_in, c.cg.n_out, c.cg.up, c.cg.down, c.cg.up_pow2) && !use_pair_fused(c.cg))
		return false;
	// (the linear output run and the zeros behind it live in the block's own part of the array)
	if (use_pair_fused(c.cg) && c.cg.in_len + 32 > c.cg.n_out) return false;
	// (one phase per thread in the one-channel kernel; the pair kernel walks tid, tid + 256, ...)
	return w.out_step <= (use_pair_fused(c.cg) ? 2048 : 256) && w.flen <= 32 && c.cg.in_len >= 4 * w.flen;
}

void Engine::fill_conv(size_t s, ConvLaunch& L, const SrcView& src) const
{
	const StagePlan& sp = plan_.stages[s];
	const StageDev& d = dev_[s];
	const ConvGeom& g = sp.cg;
	L.up = g.up; L.down = g.down; L.fl2 = g.fl2; L.bl2 = g.bl2; L.in_len = g.in_len;
	L.rot = 0; L.fl2r = g.fl2;
	L.n_in = g.n_in; L.n_out = g.n_out;
	L.blk_stride = g.in_len;
	L.blk_offset = 0;
	// aligned 16-byte loads of sample pairs need even positions on every side of the selection
	L.vec_ok = src.cur_fmt == kPcmF64 && (src.cur == nullptr || (((size_t) src.cur & 15) == 0 &&
		(src.cur_stride & 1) == 0 && (src.cur_base & 1) == 0)) && (src.ring_stride & 1) == 0 &&
		((g.in_len / g.up) & 1) == 0 ? 1 : 0;
	L.inplace = generic_conv_two_arrays(g) || generic_conv_big(g) ? 0 : 1;
	L.work = nullptr; L.work_slots = 0;
	L.up_pow2 = g.up_pow2 ? 1 : 0;
	L.down_pow2 = g.down_pow2 ? 1 : 0;
	L.n_fwd = (int) d.fwd_radix.size();
	L.n_inv = (int) d.inv_radix.size();
	if (L.n_fwd > kMaxPasses || L.n_inv > kMaxPasses)
		throw std::runtime_error("transform plan too deep");
	for (int i = 0; i < L.n_fwd; i++) L.fwd_radix[i] = d.fwd_radix[(size_t) i];
	for (int i = 0; i < L.n_inv; i++) L.inv_radix[i] = d.inv_radix[(size_t) i];
	L.H = d.H; L.Hc = d.Hc; L.tw = d.tw; L.tw_len = d.tw_len; L.spec = d.spec; L.spec2 = d.spec2; L.hp = d.hp; L.ptw = d.ptw;
	L.t_zero = 0;
	L.nch = nchw_;
	// (short transforms: fewer threads per block -- a 64-point transform on 256 threads is four waves
	// meeting at barriers with nothing to do)
	{
		int th = 64;
		while (th < opt_.at("conv_threads") && th * 8 < std::max(g.n_in, g.n_out)) th *= 2;
		L.threads = std::min(std::min(th, opt_.at("conv_threads")), 256); // (the kernels are built for <= 256)
	}
	L.src = src;
	L.tail_ring = nullptr; L.tail_p0 = L.tail_p1 = 0;
	L.tail_flags = 0; L.tail_bf = 0; L.tail_c0 = L.tail_c1 = 0;
	if (s == 0 && opt_.at("fold_tail"))
	{
		// stage 0 on the fast path: let the kernel keep the history (see process())
		const StagePlan& sp0 = plan_.stages[0];
		L.tail_ring = dev_[0].ring_alt + (long long) ch0_ * dev_[0].ring_size;
		L.tail_p1 = sp0.m;
		L.tail_p0 = sp0.m - stage_history(0);
		if (L.tail_p0 < 0) L.tail_p0 = 0;
	}
}

// Block anchoring of a fused pair (convolver s, whole-step interpolator s + 1): blocks start S virtual samples apart,
// block 0's first fresh sample sits at virtual position off.  Constants of the object (its options included).
void Engine::fused_blocking(size_t s, long long* S_out, long long* off_out) const
{
	const StagePlan& c = plan_.stages[s];
	const StagePlan& w = plan_.stages[s + 1];
	const StageDev& dw = dev_[s + 1];
	const int in_len = c.cg.in_len, fl2c = c.cg.fl2, up = c.cg.up;
	const long long In = w.in_step;
	const bool pair_two = use_pair_two(s, nullptr);
	// Blocks start S virtual samples apart with S = in_len - (interpolator taps, rounded up to
	// the up factor): the valid ranges [k*S - fl2, k*S - fl2 + in_len) of consecutive blocks
	// overlap by at least flen-1 convolver outputs, so each interpolator tap window lies inside
	// one block.  An output belongs to the FIRST block that contains its window.  That block's
	// input is complete whenever the reference has emitted the output (its latency covers one
	// whole block of in_len >= S samples), so block contents -- and therefore the stream -- do
	// not depend on how the input is cut into calls.
	long long S = in_len - (w.flen + up - 1) / up * up;
	// keep block starts on even input positions (pairs of samples load as 16 bytes)
	while ((S / up) & 1) S -= up;
	// Two-phase pair kernel: whole output GROUPS per block.  A group is Out consecutive outputs (In convolver
	// samples); the kernel's threads own phase pairs and take the groups of a block in nsets interleaved sets.
	// With S = G In and block 0 shifted by `off` so that (end of a block's valid run - right half of the
	// window) is a multiple of In, every block owns exactly G whole groups starting at phase 0: no partly
	// filled first / last group, and with G a multiple of nsets every set takes G / nsets groups (cfg2: 18
	// groups, 3 sets: 6 rounds per thread instead of 7 for 0.4 % more blocks).  Blocks stay anchored at
	// absolute positions (k S + off), so chunk invariance is untouched.  Only taken when it costs < 3 % of
	// the block's valid run.
	long long off = 0;
	if (pair_two && opt_.at("align_groups"))
	{
		long long G = S / In;
		while (G > 0 && (G * In) % up != 0) G--;
		const long long G2 = G - G % dw.nsets;
		if (G2 > 0 && (G2 * In) % up == 0 && G2 * In * 100 >= S * 98) G = G2;
		if (G > 0 && G * In * 100 >= S * 97)
		{
			// (chains with a fractional latency: the canonical stream sits fused_shift().d convolver outputs later)
			const long long r = (((long long) in_len - fl2c - w.fl2 - fused_shift(s).d) % In + In) % In;
			long long o = -r;
			for (int t = 0; t < up && o % up != 0; t++) o -= In;
			if (o % up == 0)
			{
				S = G * In;
				off = o;
			}
		}
	}
	*S_out = S;
	*off_out = off;
}

static long long ceil_div_nonneg(long long a, long long b) { return a <= 0 ? 0 : (a + b - 1) / b; }

void Engine::launch_fused(size_t s, long long wa, long long wb, const SrcView& src,
	const DstView& dst_in, void* stream)
{
	const StagePlan& c = plan_.stages[s];
	const StagePlan& w = plan_.stages[s + 1];
	// Chains with a fractional latency (fused_shift): everything below works on the CANONICAL stream -- output J at
	// position J In, phase J In mod Out, its window starting at convolver output floor(J In / Out) + D -- of which the
	// call's emitted outputs [wa, wb) are J - js; they are stored js columns (ring slots) earlier.  (js = D = 0 in
	// linear-phase chains.)
	const FusedShift fs = fused_shift(s);
	const long long D = fs.d;
	wa += fs.js;
	wb += fs.js;
	DstView dst = dst_in;
	dst.off -= fs.js;
	ConvxLaunch X;
	fill_conv(s, X.c, src);
	X.c.t_zero = fs.t_zero;
	X.c.a = 0; X.c.b = 0;
	X.c.dst = dst; // unused in fused mode
	X.in_step = w.in_step; X.out_step = w.out_step; X.flen = w.flen;
	X.fl2w = w.fl2; X.fllw = w.fll;
	X.table = dev_[s + 1].table;
	X.wtab = dev_[s + 1].wtab;
	X.wa = wa; X.wb = wb;
	X.wdst = dst;
	const int in_len = c.cg.in_len, fl2c = c.cg.fl2, up = c.cg.up;
	const long long In = w.in_step, Out = w.out_step;
	X.c.blk_offset = 0;
	const StageDev& dw = dev_[s + 1];
	int run_off = 0;
	const bool pair_two = use_pair_two(s, &run_off);
	X.run_off = run_off;
	X.ptab = dw.ptab; X.ctab = dw.ctab; X.nsets = dw.nsets;
	long long S = 0, off = 0;
	fused_blocking(s, &S, &off);
	X.c.blk_stride = (int) S;
	X.c.blk_offset = (int) off;
	if (((S / up) & 1) != 0 || ((off / up) & 1) != 0) X.c.vec_ok = 0;
	auto owner = [&](long long j) // first block whose valid range ends after the window of j
	{
		const long long v = j * In / Out + D + w.fl2 + fl2c - in_len - off;
		return v < 0 ? 0 : v / S + 1;
	};
	// outputs of block k, not clipped to the call: [first output whose window block k - 1 does not hold, first one
	// block k does not hold either)
	auto block_jlo = [&](long long k)
	{
		return k == 0 ? 0 : ceil_div_nonneg(((k - 1) * S + off - fl2c + in_len - w.fl2 - D) * Out, In);
	};
	auto block_jhi = [&](long long k) { return ceil_div_nonneg((k * S + off - fl2c + in_len - w.fl2 - D) * Out, In); };
	X.park_n = 0; X.park_out = 0; X.park_slices = 0; X.park_j0 = 0; X.park_stride = 0;
	X.walk = 0;
	X.quad = 0; X.half = 0;
	X.half_fused = opt_.at("half_fused") == 2 || (opt_.at("half_fused") == 1 && half_worth(s)) ? 1 : 0;
	X.park_src = nullptr; X.park_dst = nullptr;
	X.park_blk = SpanInfo();
	// Parked outputs (ConvxLaunch::park_*): the block that holds the call's last output is computed ONCE -- what it
	// holds beyond wb waits in the park buffer for the next call(s) instead of being computed again there (one block
	// in 13.4 for BASELINE's cfg2 call, one in 7.5 for cfg3).  ja: the first output this call has to compute.
	StageDev& dp = dev_[s];
	// (the one-channel kernel fused with the interpolator -- 16384-point blocks -- at the end of a chain: an output ring
	// of the stage's own and a copy, as in launch_stage)
	// (the one-channel form of the pair kernel fused with the interpolator -- use_solo_fused -- parks like the two-phase pair form)
	const bool solo_fused = use_solo_fused(s);
	const bool oring = opt_.at("park") && !use_pair_fused(c.cg) && !solo_fused && stage_parks(s) && dst.mask == -1 &&
		dst.fmt == kPcmF64;
	const bool parks = !oring && stage_parks(s) && dst.mask == -1 && dst.fmt == kPcmF64;
	// ... and in the middle of a chain the same block writes what it holds beyond the call AHEAD into the next stage's
	// ring (nobody reads it before it is due; the ring was sized for it -- Engine::Engine)
	const bool ahead = oring || (!parks && opt_.at("park") && (pair_two || !use_pair_fused(c.cg)) && dst.mask != -1 &&
		dst.fmt == kPcmF64 && s + 2 < plan_.stages.size());
	if (parks || oring) ensure_park(s);
	auto ring_to_rows = [&]()
	{
		TailLaunch T;
		T.src.ring = dp.park[0] + (long long) ch0_ * dp.park_stride;
		T.src.ring_stride = dp.park_stride; T.src.ring_mask = dp.park_stride - 1;
		T.src.cur = T.src.ring; T.src.cur_stride = 0; T.src.cur_base = LLONG_MAX;
		T.src.cur_fmt = kPcmF64;
		T.p0 = wa; T.p1 = wb;
		T.ring = dst.p + dst.off; T.ring_stride = dst.stride; T.ring_mask = -1;
		T.nch = nchw_;
		launch_tail(T, stream);
	};
	if (oring)
	{
		X.wdst.p = dp.park[0] + (long long) ch0_ * dp.park_stride;
		X.wdst.stride = dp.park_stride;
		X.wdst.mask = dp.park_stride - 1;
		X.wdst.off = 0;
	}
	long long ja = wa;
	if ((parks || ahead) && dp.park_end > wa)
	{
		ja = std::min(dp.park_end, wb);
		if (parks)
		{
			if (dp.park_base > wa) throw std::logic_error("parked outputs start behind the call's first output");
			X.park_src = dp.park[dp.park_cur] + (long long) ch0_ * dp.park_stride + (wa - dp.park_base);
			X.park_stride = dp.park_stride;
			X.park_j0 = wa;
			X.park_n = (int) (ja - wa);
			if (ch0_ == 0) stat_["park_calls"]++;
		}
	}
	if (ja >= wb && ahead)
	{
		// (everything this call owes is in the ring already)
		if (oring)
		{
			ring_to_rows();
			if (ch0_ == 0) stat_["park_only_calls"]++;
		}
		return;
	}
	if (ja >= wb)
	{
		// the whole call comes out of the park buffer (a short call): a plain copy; the stream's history is kept by
		// process() (tail_done_ stays false)
		TailLaunch T;
		T.src.ring = X.park_src; T.src.ring_stride = 0; T.src.ring_mask = 0;
		T.src.cur = X.park_src; T.src.cur_stride = X.park_stride; T.src.cur_base = wa;
		T.src.cur_fmt = kPcmF64;
		T.p0 = wa; T.p1 = wb;
		T.ring = dst.p + dst.off; T.ring_stride = dst.stride; T.ring_mask = -1;
		T.nch = nchw_;
		launch_tail(T, stream);
		if (ch0_ == 0) stat_["park_only_calls"]++;
		return;
	}
	const long long kfirst = owner(ja), klast = owner(wb - 1);
	// (with parked outputs the next call's first block is the one behind this call's last: whatever that one holds
	// beyond wb is parked below)
	const long long knext = parks || ahead ? klast + 1 : owner(wb);
	// (written ahead: the last block's outputs are not cut at the call's end)
	const long long wcut = ahead ? block_jhi(klast) : wb;
	if (oring ? wcut - wa > dp.park_stride :
		ahead && (wcut < wb || dst.mask + 1 < stage_history(s + 2) + plan_.stage_max_in[s + 2] + (wcut - wb)))
		throw std::logic_error("ring too small for a block written ahead");
	if (X.c.tail_ring != nullptr && (pair_two || solo_fused) && c.cg.up_pow2)
	{
		// History for the next call, exactly: its first block is knext -- the first block whose outputs this call
		// has not produced --, and no later block reads further back than that block's window (r8b_convp.h cp_load:
		// n_in input samples ending in_len / up behind the block's start).  fill_conv asked for history(), the bound
		// over every way of cutting the stream into calls: about twice what a call of MaxInLen samples needs.
		// (up is 1 or 2 here: use_pair_fused)
		if (up > 2) throw std::logic_error("fused pair kernel: up-sampling factor");
		const long long wstart = ((knext * S + off) >> (up > 1 ? 1 : 0)) - ((long long) c.cg.n_in - in_len / up);
		const long long p0 = std::min(std::max(X.c.tail_p0, wstart - 8), X.c.tail_p1);
		X.c.tail_p0 = p0 & ~1LL; // (even: pairs of samples)
		if (X.c.tail_p0 < 0) X.c.tail_p0 = 0;
	}
	double* const tail_ring = X.c.tail_ring;
	const double* const park_src = X.park_src;
	const int park_n = X.park_n;
	long long park_b = wb; // end of what the call's last block holds
	for (long long k0 = kfirst; k0 <= klast; k0 += kConvxMaxBlocks)
	{
		const long long k1 = std::min(klast, k0 + kConvxMaxBlocks - 1);
		// (the history with the call's LAST launch: its blocks are the ones that hold the tail in registers)
		X.c.tail_ring = k1 == klast ? tail_ring : nullptr;
		// (the parked outputs of the previous call with the FIRST launch)
		X.park_src = k0 == kfirst ? park_src : nullptr;
		X.park_n = k0 == kfirst ? park_n : 0;
		X.c.k0 = k0;
		X.c.nblk = (int) (k1 - k0 + 1);
		// Walk form: enough channel pairs to fill the chip with one workgroup each (2 per CU on 256 CUs: from 128 pairs on
		// a launch is worth it) and at least two blocks to walk; the launcher ignores it where the kernel has no walk form
		X.walk = 0;
		if (pair_two && (opt_.at("walk") == 2 || (opt_.at("walk") == 1 && nchw_ >= 256 && X.c.nblk >= 2)))
			X.walk = opt_.at("walk_len") > 0 ? std::min(opt_.at("walk_len"), X.c.nblk) : X.c.nblk;
		if (ch0_ == 0) stat_["conv_blocks"] += X.c.nblk;
		for (int i = 0; i < X.c.nblk; i++)
		{
			const long long k = k0 + i;
			const long long t0 = k * S + off - fl2c;    // first valid time of block k
			SpanInfo& B = X.blk[i];
			long long jlo = block_jlo(k);
			long long jhi = block_jhi(k);
			if (jlo < ja) jlo = ja;
			if (jhi > wcut) jhi = wcut;
			if (jhi < jlo) jhi = jlo;
			B.jlo = jlo; B.jhi = jhi;
			B.jlo_mod = (int) (jlo % Out);
			B.ph_lo = (int) ((jlo * In) % Out);
			B.u_lo = (int) (jlo * In / Out + D - w.fll - t0);
			B.pad = 0;
		}
		if (solo_fused && parks && k1 == klast && block_jhi(klast) > wb)
		{
			// what the call's last block holds beyond the call: [wb, end of the block) into the other park buffer (one phase
			// per thread: the span in the form of the loop above)
			park_b = block_jhi(klast);
			if (park_b - wb > dp.park_stride) throw std::logic_error("park buffer too small");
			SpanInfo& P = X.park_blk;
			P.jlo = wb; P.jhi = park_b;
			P.jlo_mod = (int) (wb % Out);
			P.ph_lo = (int) ((wb * In) % Out);
			P.u_lo = (int) (wb * In / Out + D - w.fll - (klast * S + off - fl2c));
			P.pad = 0;
			X.park_out = 1;
			X.park_dst = dp.park[dp.park_cur ^ 1] + (long long) ch0_ * dp.park_stride;
			X.park_stride = dp.park_stride;
		}
		if (pair_two)
		{
			auto two_phase_span = [&](SpanInfo& B, long long k)
			{
				B.pad = 0;
				if (B.jhi <= B.jlo) return;
				const long long t0 = k * S + off - fl2c;
				const long long g0 = B.jlo / Out, glast = (B.jhi - 1) / Out;
				B.ph_lo = (int) (glast - g0);
				B.pad = (int) (B.jhi - glast * Out); // the phase the block's last group ends before
				B.u_lo = (int) (In * g0 + D - w.fll - t0) + run_off;
			};
			for (int i = 0; i < X.c.nblk; i++) two_phase_span(X.blk[i], k0 + i);
			if (parks && k1 == klast && block_jhi(klast) > wb)
			{
				// what the call's last block holds beyond the call: [wb, end of the block) into the other park buffer
				park_b = block_jhi(klast);
				if (park_b - wb > dp.park_stride) throw std::logic_error("park buffer too small");
				SpanInfo& P = X.park_blk;
				P.jlo = wb; P.jhi = park_b;
				P.jlo_mod = (int) (wb % Out);
				P.ph_lo = 0; P.u_lo = 0;
				two_phase_span(P, klast);
				X.park_out = 1;
				X.park_dst = dp.park[dp.park_cur ^ 1] + (long long) ch0_ * dp.park_stride;
				X.park_stride = dp.park_stride;
			}
			// (blocks the launcher put on the walk body: counted per engine, once per call like conv_blocks)
			const long long w0 = launch_walk_blocks();
			launch_convp(X, (dw.taps2 == 27 ? 5 : 4) + (c.cg.complex_h ? 12 : 0), stream);
			if (ch0_ == 0) stat_["walk_blocks"] += launch_walk_blocks() - w0;
		}
		else if (c.cg.complex_h)
			// (fuse_latency_ok admits a complex spectrum only where the two-phase tables exist)
			throw std::logic_error("fused launch: complex kernel spectrum without the two-phase tables");
		else if (use_pair_fused(c.cg)) launch_convp(X, 1, stream);
		else if (use_solo_fused(s)) launch_convp(X, 18, stream);
		else launch_convx(X, 1, stream);
		if (X.c.tail_ring != nullptr) tail_done_ = true;
	}
	if (oring) ring_to_rows();
	if ((parks || ahead) && ch0_ + nchw_ >= nch_)
	{
		// (the counters once per call, after its last channel window)
		if (ahead) park_b = wcut;
		if (parks && park_b > wb) dp.park_cur ^= 1;
		dp.park_base = wb;
		dp.park_end = park_b;
	}
}

} // namespace r8bhip
