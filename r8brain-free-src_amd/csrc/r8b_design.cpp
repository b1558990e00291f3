// r8b_design.cpp -- see r8b_design.h.  Host only; compiled with -ffp-contract=off so that the
// transcendental-heavy fits round the way a plain x86-64 build of the reference does.
#include "r8b_design.h"

#include <cmath>
#include <complex>
#include <list>
#include <map>
#include <mutex>
#include <tuple>

#include "r8b_tables.inc"

namespace r8bhip {

namespace {

const double kPi = 3.14159265358979324;

// The reference's window uses the Abramowitz-Stegun polynomial, not the true I0
// (reference r8bbase.h:1192-1212); the filters depend on its exact values.
double bessel_i0_as(double x)
{
	const double ax = std::fabs(x);
	if (ax < 3.75)
	{
		double y = x / 3.75;
		y *= y;
		return 1.0 + y * (3.5156229 + y * (3.0899424 + y * (1.2067492 +
			y * (0.2659732 + y * (0.360768e-1 + y * 0.45813e-2)))));
	}
	const double y = 3.75 / ax;
	return std::exp(ax) / std::sqrt(ax) * (0.39894228 + y * (0.1328592e-1 +
		y * (0.225319e-2 + y * (-0.157565e-2 + y * (0.916281e-2 +
		y * (-0.2057706e-1 + y * (0.2635537e-1 + y * (-0.1647633e-1 +
		y * 0.392377e-2))))))));
}

// reference r8bbase.h:1154-1157
inline double pow_abs(double v, double p) { return std::exp(p * std::log(std::fabs(v) + 1e-300)); }

// reference r8bbase.h:1176-1179 (its own asinh, cancellation included)
inline double asinh_plain(double v) { return std::log(v + std::sqrt(v * v + 1.0)); }

// Kaiser window raised to a power, evaluated sample by sample the way
// CDSPSincFilterGen::calcWindowKaiser does (reference CDSPSincFilterGen.h:230-241, 586-605).
struct KaiserPow
{
	double beta, mul, len2i, frac, power;
	int wn;
	KaiserPow(double len2, double b, double pw, bool centered, int fl2, double frac_delay)
	{
		beta = b < 1.0 ? 1.0 : (b > 350.0 ? 350.0 : b);
		mul = 1.0 / bessel_i0_as(beta);
		len2i = 1.0 / len2;
		frac = frac_delay * len2i;
		power = std::fabs(pw);
		wn = centered ? 0 : -fl2;
	}
	double next_raw()
	{
		const double a = wn * len2i + frac;
		const double n = 1.0 - a * a;
		wn++;
		if (n <= 0.0) return 0.0;
		return bessel_i0_as(beta * std::sqrt(n)) * mul;
	}
	double next() { return pow_abs(next_raw(), power); }
};

// (pwr, hl, fo1) of the low-pass fits, reference CDSPFIRFilter.h:222-448.
void lp_fit(double trans_band, double atten_req, double* pwr_o, double* hl_o, double* fo1_o)
{
	const double tb = trans_band * 0.01;
	double atten = -atten_req;
	const int band = tb >= 0.25 ? 0 : (tb >= 0.10 ? 1 : 2);
	const int lvl = atten_req >= 117.0 ? 0 : (atten_req >= 60.0 ? 1 : 2);
	static const double first_corr[3][3] = {
		{ 1.60, 1.91, 2.25 }, { 0.69, 0.73, 1.13 }, { 0.21, 0.25, 0.36 } };
	atten -= first_corr[band][lvl];
	int idx = (int) std::floor((-atten - 49.0) * 264 / 176.25 + 0.5);
	if (idx < 0) idx = 0;
	if (idx > 264) idx = 264;
	const signed char* corr = band == 0 ? kAttenCorr0 : (band == 1 ? kAttenCorr1 : kAttenCorr2);
	const double scale = band == 0 ? kAttenCorrScale0 :
		(band == 1 ? kAttenCorrScale1 : kAttenCorrScale2);
	atten -= corr[idx] / scale;
	const double a = atten;
	using std::cos; using std::sin; using std::tan; using std::atan; using std::atan2;
	using std::exp; using std::sqrt; using std::tanh; using std::cosh; using std::log;
	using std::pow;
	const double pwr = 7.43932822146293e-8 * a * a + 0.000102747434588003 *
		cos(0.00785021930010397 * a) * cos(0.633854318781239 + 0.103208573657699 * a) -
		0.00798132247867036 - 0.000903555213543865 * a - 0.0969365532127236 *
		exp(0.0779275237937911 * a) - 1.37304948662012e-5 * a * cos(0.00785021930010397 * a);
	double hl, fo1;
	if (pwr <= 0.067665322581)
	{
		if (band == 0)
		{
			hl = 2.6778150875894 / tb + 300.547590563091 * atan(atan(
				2.68959772209918 * pwr)) / (5.5099277187035 * tb - tb * tanh(cos(asinh_plain(a))));
			fo1 = 0.987205355829873 * tb + 1.00011788929851 * atan2(
				-0.321432067051302 - 6.19131357321578 * sqrt(pwr),
				hl + -1.14861472207245 / (hl - 14.1821147585957) + pow(0.9521145021664,
				pow(atan2(1.12018764830637, tb), 2.10988901686912 * hl - 20.9691278378345)));
		}
		else if (band == 1)
		{
			hl = (1.56688617018066 + 142.064321294568 * pwr + 0.00419441117131136 * cos(
				243.633511747297 * pwr) - 0.022953443903576 * a - 0.026629568860284 * cos(
				127.715550622571 * pwr)) / tb;
			fo1 = 0.982299356642411 * tb + 0.999441744774215 * asinh_plain(
				(-0.361783054039583 - 5.80540593623676 * sqrt(pwr)) / hl);
		}
		else
		{
			hl = (2.45739657014937 + 269.183679500541 * pwr * cos(5.73225668178813 + atan2(
				cosh(0.988861169868941 - 17.2201556280744 * pwr), 1.08340138240431 * pwr))) / tb;
			fo1 = 2.291956939 * tb + 0.01942450693 * tb * tb * hl - 4.67538973161837 * pwr * tb -
				1.668433124 * tb * pow(pwr, pwr);
		}
	}
	else
	{
		if (band == 0)
		{
			hl = (1.50258368698213 + 158.556968859477 * asinh_plain(pwr) * tanh(
				57.9466246871383 * tanh(pwr)) - 0.0105440479814834 * a) / tb;
			fo1 = 0.994024401639321 * tb + (-0.236282717577215 - 6.8724924545387 * sqrt(
				sin(pwr))) / hl;
		}
		else if (band == 1)
		{
			hl = (1.50277377248945 + 158.222625721046 * asinh_plain(pwr) * tanh(
				1.02875299001715 + 42.072277322604 * pwr) - 0.0108380943845632 * a) / tb;
			fo1 = 0.992539376734551 * tb + (-0.251747813037178 - 6.74159892452584 * sqrt(
				tanh(tanh(tan(pwr))))) / hl;
		}
		else
		{
			hl = (1.15990238966306 * pwr - 5.02124037125213 * pwr * pwr - 0.158676856669827 * a *
				cos(1.1609073390614 * pwr - 6.33932586197475 * pwr * pwr * pwr)) / tb;
			fo1 = 0.867344453126885 * tb + 0.052693817907757 * tb * log(pwr) +
				0.0895511178735932 * tb * atan(59.7538527741309 * pwr) -
				0.0745653568081453 * pwr * tb;
		}
	}
	*pwr_o = pwr;
	*hl_o = hl;
	*fo1_o = fo1;
}

std::mutex g_cache_mutex;

// Bounded cache in the manner of the reference's (CDSPFIRFilter.h:598-694): most recently used first; when a new
// entry finds the cache full, entries nobody else holds go, oldest first -- the count exceeds the bound only while
// that many different objects are in use at the same time.  (Callers hold g_cache_mutex.)
template<class Key, class T>
struct LruCache
{
	typedef std::shared_ptr<const T> Ref;
	std::list<std::pair<Key, Ref>> items;
	size_t max;
	explicit LruCache(size_t m) : max(m) {}
	Ref find(const Key& k)
	{
		for (auto it = items.begin(); it != items.end(); ++it)
			if (it->first == k)
			{
				items.splice(items.begin(), items, it);
				return items.front().second;
			}
		return Ref();
	}
	Ref insert(const Key& k, T&& v)
	{
		for (auto it = items.end(); items.size() >= max && it != items.begin();)
		{
			--it;
			if (it->second.use_count() == 1) it = items.erase(it);
		}
		items.emplace_front(k, std::make_shared<const T>(std::move(v)));
		return items.front().second;
	}
};

typedef std::tuple<double, double, double, double, bool, int> LpKey;
typedef std::tuple<int, int, int, double, bool> BankKey;
LruCache<LpKey, LpFilter>& lp_cache()
{
	static LruCache<LpKey, LpFilter> c((size_t) kFilterCacheMax);
	return c;
}
LruCache<BankKey, FracBank>& bank_cache()
{
	static LruCache<BankKey, FracBank> c((size_t) kFracBankCacheMax);
	return c;
}

} // namespace

namespace { int g_lane_deals = 0; }
void lane_deal_cache_count(int delta_or_zero, int* count)
{
	std::lock_guard<std::mutex> lock(g_cache_mutex);
	g_lane_deals += delta_or_zero;
	if (count) *count = g_lane_deals;
}

void design_cache_counts(int counts[3])
{
	std::lock_guard<std::mutex> lock(g_cache_mutex);
	counts[0] = (int) lp_cache().items.size();
	counts[1] = (int) bank_cache().items.size();
	counts[2] = g_lane_deals;
}

namespace {

// in-place complex FFT (iterative radix 2), sign = -1: forward
template<class T>
void fft_host(std::vector<std::complex<T>>& a, int sign)
{
	const size_t n = a.size();
	for (size_t i = 1, j = 0; i < n; i++)
	{
		size_t bit = n >> 1;
		for (; j & bit; bit >>= 1) j ^= bit;
		j ^= bit;
		if (i < j) std::swap(a[i], a[j]);
	}
	const long double two_pi = 6.283185307179586476925286766559L;
	for (size_t len = 2; len <= n; len <<= 1)
	{
		std::vector<std::complex<T>> w(len / 2);
		for (size_t k = 0; k < len / 2; k++)
			w[k] = std::complex<T>((T) cosl(two_pi * k / len), (T) (sign * sinl(two_pi * k / len)));
		for (size_t i = 0; i < n; i += len)
			for (size_t k = 0; k < len / 2; k++)
			{
				const std::complex<T> u = a[i + k], v = a[i + k + len / 2] * w[k];
				a[i + k] = u + v;
				a[i + k + len / 2] = u - v;
			}
	}
}

// Minimum-phase transform of a FIR kernel through the cepstrum (reference CDSPRealFFT.h:681-785,
// LenMult = 16, DoFinalMul = false): log-magnitude spectrum -> cepstrum -> discrete Hilbert transform
// (positive quefrencies kept, negative ones negated, 0 and N/2 cleared) -> phase -> spectrum with the
// original magnitudes -> time domain, first kernel_len samples.  The scale of the result is left as
// it comes (the caller normalises the DC gain).
// Arithmetic: fp64 like the reference, on purpose.  The deep stop band of these kernels (-180 dB and
// below) lies under the rounding noise of an fp64 transform of this length; log|H| there -- and with
// it the phase the Hilbert transform derives -- is set by that noise floor, so a more accurate
// transform gives a (truer but) different filter: in extended precision the 1/3-band kernel comes
// out 0.011 samples earlier than the reference's.
template<class T>
std::vector<double> min_phase_transform_t(const std::vector<double>& kernel)
{
	typedef std::complex<T> C;
	const int klen = (int) kernel.size();
	const int bits = bit_occupancy((long long) klen * 16 - 1);
	const size_t n = (size_t) 1 << bits, n2 = n / 2;
	std::vector<C> a(n, C(0, 0));
	for (int i = 0; i < klen; i++) a[(size_t) i] = C((T) kernel[(size_t) i], 0);
	fft_host(a, -1);
	std::vector<T> mag(n2 + 1);
	const T x0 = a[0].real(), xn = a[n2].real();
	for (size_t k = 0; k <= n2; k++) mag[k] = std::sqrt(a[k].real() * a[k].real() + a[k].imag() * a[k].imag());
	mag[0] = std::abs(x0);
	mag[n2] = std::abs(xn);
	std::vector<C> l(n);
	for (size_t k = 0; k <= n2; k++)
	{
		const T v = std::log(mag[k] + (T) 1e-300);
		l[k] = C(v, 0);
		if (k != 0 && k != n2) l[n - k] = C(v, 0);
	}
	fft_host(l, +1); // cepstrum * n
	std::vector<C> d(n);
	d[0] = d[n2] = C(0, 0);
	for (size_t i = 1; i < n2; i++) d[i] = C(l[i].real() / (T) n, 0);
	for (size_t i = n2 + 1; i < n; i++) d[i] = C(-l[i].real() / (T) n, 0);
	fft_host(d, -1); // purely imaginary: the phase
	std::vector<C> s(n);
	s[0] = C(x0, 0);
	s[n2] = C(xn, 0);
	for (size_t k = 1; k < n2; k++)
	{
		const T ph = d[k].imag();
		s[k] = C(std::cos(ph) * mag[k], std::sin(ph) * mag[k]);
		s[n - k] = std::conj(s[k]);
	}
	fft_host(s, +1);
	std::vector<double> out((size_t) klen);
	for (int i = 0; i < klen; i++) out[(size_t) i] = (double) (s[(size_t) i].real() / (T) n);
	return out;
}

std::vector<double> min_phase_transform(const std::vector<double>& kernel)
{
	return min_phase_transform_t<double>(kernel);
}

} // namespace

#ifdef R8B_TEST_HOOKS
// (test builds only -- tests/emul and the GPU tier's libr8bsrc_hip_testhooks.so; the shipped library does not
// contain the hook: r8b_design.h)
namespace {
LpProvider g_lp_provider = nullptr;
int g_lp_provider_gen = 0; // filters made under a provider are cached apart from the designer's own
}

void set_lp_provider(LpProvider p)
{
	std::lock_guard<std::mutex> lock(g_cache_mutex);
	static int installs = 0;
	g_lp_provider = p;
	g_lp_provider_gen = p != nullptr ? ++installs : 0;
}
#endif

LpFilterRef design_lp(double norm_freq, double trans_band, double atten, double gain, bool min_phase)
{
	typedef LpKey Key;
	LruCache<LpKey, LpFilter>& cache = lp_cache();
	std::unique_lock<std::mutex> lock(g_cache_mutex);
	int gen = 0;
#ifdef R8B_TEST_HOOKS
	const LpProvider provider = g_lp_provider;
	gen = provider != nullptr ? g_lp_provider_gen : 0;
#endif
	const Key key(norm_freq, trans_band, atten, gain, min_phase, gen);
	if (LpFilterRef hit = cache.find(key)) return hit;
#ifdef R8B_TEST_HOOKS
	if (provider != nullptr)
	{
		// the provider runs with the cache unlocked (it may call back into r8b_design_*); the cache is looked up
		// again afterwards
		lock.unlock();
		std::vector<double> t((size_t) 1 << 18);
		int lat = 0, bits = 0;
		double lf = 0.0;
		const int n = provider(norm_freq, trans_band, atten, gain, min_phase ? 1 : 0, t.data(), (int) t.size(),
			&lat, &lf, &bits);
		lock.lock();
		if (LpFilterRef hit = cache.find(key)) return hit;
		if (n > 0)
		{
			LpFilter f;
			f.taps.assign(t.begin(), t.begin() + n);
			f.kernel_len = n;
			f.fl2 = lat;
			f.lat_frac = lf;
			f.zero_phase = !min_phase;
			f.block_len_bits = bits;
			return cache.insert(key, std::move(f));
		}
	}
#endif

	double pwr, hl, fo1;
	lp_fit(trans_band, atten, &pwr, &hl, &fo1);
	const double len2 = 0.25 * hl / norm_freq;
	const double freq2 = kPi * (1.0 - fo1) * norm_freq;
	const int fl2 = (int) std::floor(len2);

	LpFilter f;
	f.fl2 = fl2;
	f.kernel_len = 2 * fl2 + 1;
	f.block_len_bits = bit_occupancy(f.kernel_len - 1);
	f.taps.assign((size_t) f.kernel_len, 0.0);

	// windowed sinc, sine by the 2-term recurrence the reference uses
	// (reference r8bbase.h:741-749, CDSPSincFilterGen.h:315-338)
	KaiserPow win(len2, 125.0, pwr, true, fl2, 0.0);
	double s1 = std::sin(0.0) * (1.0 / kPi);
	double s2 = std::sin(0.0 - freq2) * (1.0 / kPi);
	const double incr = 2.0 * std::cos(freq2);
	{
		const double r = s1; // first generate() call is discarded by the designer
		s1 = incr * r - s2;
		s2 = r;
	}
	double* c = &f.taps[(size_t) fl2];
	c[0] = freq2 * win.next() / kPi;
	for (int t = 1; t <= fl2; t++)
	{
		const double r = s1;
		s1 = incr * r - s2;
		s2 = r;
		const double v = r * win.next() / t;
		c[t] = v;
		c[-t] = v;
	}
	if (min_phase)
	{
		// reference CDSPFIRFilter.h:476-484, 520-528: transform the raw kernel, take the group delay at
		// DC as the latency (integer part consumed by the convolver, fraction handed to the next stage),
		// then normalise.  The reference estimates the delay by a finite difference of the phase at
		// 1e-9 rad (r8bbase.h:876-920); d(phase)/d(omega) at 0 is sum(n h[n]) / sum(h[n]) exactly.
		f.taps = min_phase_transform(f.taps);
		long double sn = 0.0L, sh = 0.0L;
		for (int i = 0; i < f.kernel_len; i++)
		{
			sn += (long double) i * f.taps[(size_t) i];
			sh += f.taps[(size_t) i];
		}
		const double gd = (double) (sn / sh);
		f.zero_phase = false;
		f.fl2 = (int) gd;
		f.lat_frac = gd - f.fl2;
	}
	double s = 0.0;
	for (int i = 0; i < f.kernel_len; i++) s += f.taps[(size_t) i];
	s = gain / s;
	for (int i = 0; i < f.kernel_len; i++) f.taps[(size_t) i] *= s;
	return cache.insert(key, std::move(f));
}

namespace {

// Kaiser parameters of the fractional-delay bank, reference CDSPFracInterpolator.h:279-341.
void frac_win_params(double atten, bool third, double* beta, double* power, double* att_r,
	int* flen)
{
	const double (*rows)[3] = third ? kFracCoeffs3 : kFracCoeffs2;
	const int n = third ? kFracCoeffs3Count : kFracCoeffs2Count;
	const int base = third ? kFracCoeffs3Base : kFracCoeffs2Base;
	int i = 0;
	while (i != n - 1 && rows[i][2] < atten) i++;
	*beta = rows[i][0];
	*power = rows[i][1];
	*att_r = rows[i][2];
	*flen = base + 2 * i;
}

// One fractional-delay filter: reference CDSPSincFilterGen.h:168-177 (initFrac) and 452-552
// (generateFrac, power branch), then unit-DC-gain normalisation (r8bbase.h:931-961).
void frac_filter(double fd, int flen, double beta, double power, double* op, int opinc)
{
	const double len2 = (double) (flen / 2);
	const int fl2 = (int) std::ceil(len2);
	KaiserPow win(len2, beta, power, false, fl2, fd);
	double* p = op;
	int t = -fl2;
	if (t + fd < -len2)
	{
		win.next_raw();
		*p = 0.0;
		p += opinc;
		t++;
	}
	double f = std::sin(fd * kPi) / kPi;
	if ((t & 1) != 0) f = -f;
	int is_zero_x = std::fabs(fd - 1.0) < 2.3e-13 ? 1 : 0;
	int mt = 0 - is_zero_x;
	is_zero_x = (is_zero_x || std::fabs(fd) < 2.3e-13) ? 1 : 0;
	while (t < mt)
	{
		*p = f * win.next() / (t + fd);
		p += opinc;
		t++;
		f = -f;
	}
	if (is_zero_x) *p = win.next();
	else *p = f * win.next() / fd;
	mt = fl2 - 2;
	while (t < mt)
	{
		p += opinc;
		t++;
		f = -f;
		*p = f * win.next() / (t + fd);
	}
	p += opinc;
	t++;
	f = -f;
	const double ut = t + fd;
	*p = ut > len2 ? 0.0 : f * win.next() / ut;

	double s = 0.0;
	for (int i = 0; i < flen; i++) s += op[(size_t) i * opinc];
	s = 1.0 / s;
	for (int i = 0; i < flen; i++) op[(size_t) i * opinc] *= s;
}

} // namespace

FracBankRef design_frac_bank(int fracs, int element_size, int interp_points, double atten,
	bool third)
{
	typedef BankKey Key;
	LruCache<BankKey, FracBank>& cache = bank_cache();
	std::lock_guard<std::mutex> lock(g_cache_mutex);
	const Key key(fracs, element_size, interp_points, atten, third);
	if (FracBankRef hit = cache.find(key)) return hit;

	double beta, power, att_r;
	int flen;
	frac_win_params(atten, third, &beta, &power, &att_r, &flen);
	if (fracs == -1) fracs = (int) std::ceil(std::pow(6.4, att_r / 50.0));

	FracBank b;
	b.filter_len = flen;
	b.fracs = fracs;
	b.element_size = element_size;
	b.atten = att_r;
	const int pc2 = interp_points / 2;
	const int nrows = fracs + interp_points; // bank indices -pc2+1 .. fracs+pc2
	std::vector<double> filt((size_t) nrows * flen);
	for (int r = 0; r < nrows; r++)
	{
		const int i = r - pc2 + 1;
		frac_filter((double) (fracs - i) / fracs, flen, beta, power, &filt[(size_t) r * flen], 1);
	}
	b.table.assign((size_t) (fracs + 1) * flen * element_size, 0.0);
	if (element_size == 1)
	{
		// interp_points == 2: bank index i lives in row r = i (pc2 == 1)
		for (int i = 0; i <= fracs; i++)
			for (int k = 0; k < flen; k++)
				b.table[(size_t) i * flen + k] = filt[(size_t) (i + pc2 - 1) * flen + k];
	}
	else
	{
		// 8-point, 2nd-order spline over bank indices i-3..i+4 (reference r8bbase.h:1014-1024):
		// row r of `filt` is bank index r-3, so index i reads rows i..i+7 with x0 = row i+3.
		const double k = 1.31578947368421052e-2;
		for (int i = 0; i <= fracs; i++)
			for (int t = 0; t < flen; t++)
			{
				const double* y = &filt[(size_t) i * flen + t];
				const double xm3 = y[0], xm2 = y[flen], xm1 = y[2 * flen], x0 = y[3 * flen],
					x1 = y[4 * flen], x2 = y[5 * flen], x3 = y[6 * flen], x4 = y[7 * flen];
				double* o = &b.table[((size_t) i * flen + t) * 3];
				o[0] = x0;
				o[1] = (61.0 * (x1 - xm1) + 16.0 * (xm2 - x2) + 3.0 * (x3 - xm3)) * k;
				o[2] = (106.0 * (xm1 + x1) + 10.0 * x3 + 6.0 * xm3 - 3.0 * x4 -
					29.0 * (xm2 + x2) - 167.0 * x0) * k;
			}
	}
	return cache.insert(key, std::move(b));
}

int select_hb_filter(double atten, int steep, bool third, const double** taps, double* att)
{
	if (steep < 0) steep = 0;
	if (steep > 6) steep = 6;
	const int first = third ? kHB_third_steep[steep][0] : kHB_half_steep[steep][0];
	const int count = third ? kHB_third_steep[steep][1] : kHB_half_steep[steep][1];
	int k = 0;
	if (third)
	{
		while (k != count - 1 && kHB_third_rows[first + k].att < atten) k++;
		*taps = &kHB_third_taps[kHB_third_rows[first + k].offs];
		if (att) *att = kHB_third_rows[first + k].att;
		return kHB_third_rows[first + k].ntaps;
	}
	while (k != count - 1 && kHB_half_rows[first + k].att < atten) k++;
	*taps = &kHB_half_taps[kHB_half_rows[first + k].offs];
	if (att) *att = kHB_half_rows[first + k].att;
	return kHB_half_rows[first + k].ntaps;
}

// reference CDSPFracInterpolator.h:609-673
bool whole_stepping(double ssr, double dsr, int* in_step, int* out_step)
{
	double l = ssr, s = dsr, g = -1.0;
	for (int it = 1; it < 150; it++)
	{
		const double r = l - s;
		if (r == 0.0)
		{
			g = s;
			break;
		}
		l = s;
		s = std::fabs(r);
	}
	*in_step = 0;
	*out_step = 0;
	if (!(g > 0.0)) return false;
	const double i0 = ssr / g, o0 = dsr / g;
	*in_step = (int) i0;
	*out_step = (int) o0;
	if (i0 != (double) (int) i0 || o0 != (double) (int) o0) return false;
	if ((int) o0 > 1500) return false;
	return true;
}

std::vector<StageDesc> build_topology(double src, double dst, double tb, double atten, int phase)
{
	std::vector<StageDesc> st;
	auto conv = [&](double nf, double tbv, double gain, int up, int down)
	{
		StageDesc d;
		d.kind = kConv; d.a = nf; d.b = tbv; d.c = atten; d.d = gain; d.i0 = up; d.i1 = down;
		d.phase = phase;
		st.push_back(d);
	};
	auto frac = [&](double s, double dd, bool third)
	{
		StageDesc d;
		d.kind = kFrac; d.a = s; d.b = dd; d.c = atten; d.i0 = third ? 1 : 0;
		st.push_back(d);
	};
	auto hb = [&](StageKind k, int steep, bool third)
	{
		StageDesc d;
		d.kind = k; d.a = atten; d.i0 = steep; d.i1 = third ? 1 : 0;
		st.push_back(d);
	};
	if (src == dst) return st;

	// fixed small ratios handled by one convolver (reference CDSPResampler.h:146-172)
	static const int common[5][2] = { { 1, 2 }, { 1, 3 }, { 2, 3 }, { 3, 2 }, { 3, 4 } };
	for (int i = 0; i < 5; i++)
	{
		const int num = common[i][0], den = common[i][1];
		if (src * num == dst * den)
		{
			conv(1.0 / (num > den ? num : den), tb, (double) num, num, den);
			return st;
		}
	}
	// 2^c and 3*2^c upsampling: convolver then half-band stages (reference :176-216)
	for (int i = 2; i <= 3; i++)
	{
		int c = 0;
		bool found = false;
		while (true)
		{
			const double nsr = src * (i << c);
			if (nsr == dst) { found = true; break; }
			if (nsr > dst) break;
			c++;
		}
		if (found)
		{
			conv(1.0 / i, tb, (double) i, i, 1);
			for (int s = 0; s < c; s++) hb(kHBUp, s, i == 3);
			return st;
		}
	}
	if (dst * 2.0 > src)
	{
		// upsampling or mild downsampling: 2x convolver + interpolator (reference :218-330)
		const double nf = dst > src ? 0.5 : 0.5 * dst / src;
		conv(nf, tb, 2.0, 2, 1);
		const double tbw = 0.0175;
		const double thresh = src / (1.0 - tbw * tb);
		int c = 0, div = 1;
		while (true)
		{
			const int nd = div * 2;
			if (dst < thresh * nd) break;
			div = nd;
			c++;
		}
		int c2 = 0, div2 = 1;
		while (true)
		{
			const int nd = div * (c2 == 0 ? 3 : 2);
			if (dst < thresh * nd) break;
			div2 = nd;
			c2++;
		}
		const double src2 = src * 2.0;
		int tmp1, tmp2;
		if (c == 1 && whole_stepping(src2, dst, &tmp1, &tmp2)) c = 0;
		if (c > 0)
		{
			int num = 2;
			if (c2 > 0 && div2 > div)
			{
				div = div2;
				c = c2;
				num = 3;
			}
			frac(src2 * div, dst, false);
			double tb2 = (1.0 - src * div / dst) / tbw;
			if (tb2 > 45.0) tb2 = 45.0;
			conv(1.0 / num, tb2, (double) num, num, 1);
			for (int s = 1; s < c; s++) hb(kHBUp, s - 1, num == 3);
		}
		else
		{
			frac(src2, dst, false);
		}
		return st;
	}
	// strong downsampling: half-band decimators, convolver, interpolator (reference :332-393)
	double check = dst * 4.0;
	int c = 0;
	double fin_gain = 1.0;
	while (check <= src)
	{
		c++;
		check *= 2.0;
		fin_gain *= 0.5;
	}
	const int srdiv = 1 << c;
	double nf = 0.5;
	bool use_interp = true, third = false;
	int downf = 1;
	for (int df = 2; df <= 3; df++)
	{
		if (dst * srdiv * df == src)
		{
			nf = 1.0 / df;
			use_interp = false;
			third = df == 3;
			downf = df;
			break;
		}
	}
	if (use_interp)
	{
		downf = 1;
		nf = dst * srdiv / src;
		third = nf * 3.0 <= 1.0;
	}
	for (int i = 0; i < c; i++) hb(kHBDown, c - 1 - i, third);
	conv(nf, tb, fin_gain, 1, downf);
	if (use_interp) frac(src, dst * srdiv, third);
	return st;
}

} // namespace r8bhip
