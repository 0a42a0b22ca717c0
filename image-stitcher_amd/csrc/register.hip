// Registration kernels for gfx950: batched phase cross-correlation of tile-overlap crops.
//
// Replaces, per pair of tiles: normalize_image (stitcher.py:613-617), the crops of
// calculate_horizontal_shift / calculate_vertical_shift (:504-506, :517-519) and
// skimage.registration.phase_cross_correlation(upsample_factor=u)
// (skimage/registration/_phase_cross_correlation.py:189-276 of 0.18.3, plus the one-line
// `normalization="phase"` of >= 0.19).  All arithmetic is float64/complex128 like the reference
// (uint16 input -> complex128 FFT).  No dense contraction worth MFMA: the work is FFT butterflies
// in LDS and two small upsampled-DFT sums.
//
// Pipeline (one launch each, batched over pairs; spectra live in the caller's workspace and stay
// L2 / Infinity-Cache resident -- 2 x n0 x (n1/2+1) complex128 per pair):
//   K0 init        twiddle tables exp(-2 pi i k/n) for both axes and exp(+2 pi i j/(n u)) for the
//                  upsampled DFT (sincospi on exactly reduced integer arguments)
//   K1 rows fwd    crop + min-max normalise + two-for-one real FFT along axis 1:
//                  z = ref_row + i mov_row, one complex FFT, split into the two half spectra
//   K2 columns     FFT along axis 0 of both half spectra (tiles of TC columns through LDS),
//                  cross-power product F conj(G) [/ max(|.|, 100 eps)], |F|^2 / |G|^2 partial sums,
//                  product stored for K4, inverse FFT along axis 0 in place
//   K3 rows inv    Hermitian-extend two rows, one inverse complex FFT -> two real correlation rows,
//                  |.| and per-block argmax (first index wins ties, NaN wins like numpy)
//   K4a peak       reduce block partials -> whole-pixel peak, wrap to signed shift (skimage :215-220)
//   K4b upsample 1 D1[k0][b] = sum_k1 P[k0][k1] e^{+2 pi i (b-off1) f1[k1]}   (skimage :63-75, last axis)
//   K4c upsample 2 cc_up[a][b] = sum_k0 D1[k0][b] e^{+2 pi i (a-off0) f0[k0]}   (16 k0-slices per block, fixed order)
//   K4d argmax |cc_up| in row-major order (skimage :244)
#include <hip/hip_runtime.h>

#include "common.h"

using namespace sq;

namespace {

struct cplx {
    double re, im;
};
// complex product with one rounding fewer per component (2 multiplies + 2 fused multiply-adds instead of 4 + 2: the
// transforms and the upsampled DFT are bound by VALU issue, and nothing here is compared bit for bit with another FFT)
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {fma(a.re, b.re, -(a.im * b.im)), fma(a.re, b.im, a.im * b.re)}; }
// acc + a * b, four fused multiply-adds
__device__ __forceinline__ cplx cfma(cplx a, cplx b, cplx acc) {
    return {fma(-a.im, b.im, fma(a.re, b.re, acc.re)), fma(a.im, b.re, fma(a.re, b.im, acc.im))};
}
// walks e = l * n + j in steps of nt without a division per element: (l, j) of the first element and of the step
struct Walk {
    int l, j, dl, dj, n;
    __device__ __forceinline__ Walk(int first, int step, int n_) : l(first / n_), j(first % n_), dl(step / n_), dj(step % n_), n(n_) {}
    __device__ __forceinline__ void next() {
        l += dl;
        j += dj;
        if (j >= n) {
            j -= n;
            ++l;
        }
    }
};
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cconj(cplx a) { return {a.re, -a.im}; }

// One line FFT runs in place in LDS (one line of complex128 per transform):
//   * a power of two: radix-2 decimation in time, two stages fused per pass (lines_fft_pow2);
//   * any other length whose prime factors are all <= 13 ("smooth": 1500 = 2^2 3 5^3, 6000 = 2^4 3 5^3, 300, 80 ...):
//     mixed-radix Cooley-Tukey with radix 4 / 2 / 3 / 5 / 7 / 11 / 13 butterflies (lines_fft_mixed), like pocketfft --
//     the reference's FFT -- does for such lengths;
//   * any other length n (large prime factors: 2084 = 4 * 521, 3122 = 2 * 7 * 223, 1031): Bluestein's chirp-z form
//     through a SMOOTH length M >= 2n - 1 (not the next power of two: 2084 -> 4320 instead of 8192 points), which is
//     how pocketfft treats those too.
// A line of up to 9728 points fits the 160 KB of LDS beside a few hundred bytes of reduction scratch: any smooth crop side up
// to 9728 and any crop side at all up to 4860 (a 9568 x 6380 sensor gives 4784 and 3190) are transformed there.
// LONGER lines (round 4; the reference's pocketfft takes any length, stitcher.py:503-510, 516-523) run the SAME transforms
// with the line in the workspace instead -- one line per workgroup in one of LONG_SLOTS scratch lines, passes separated by
// the same barriers (a workgroup's waves share their CU's L1, so a workgroup-scope barrier publishes global stores too);
// such a line lives in the L2 (a 2^17-point line is 2 MB).  Any crop side up to 65535 is accepted that way: not fast
// (flat loads and stores instead of LDS), but computed instead of refused.
constexpr int MAX_LINE = 9728;              // points of a line in LDS
constexpr int MAX_LONG_LINE = 1 << 17;      // points of a line in the workspace
constexpr int MAX_CROP_SIDE = 65535;        // (2 * 65535 - 1 <= 2^17: every such side has a Bluestein length)
constexpr int LONG_SLOTS = 512;             // scratch lines = workgroups of a long-line launch
constexpr int LONG_THREADS = 512;           // ... of 512 threads: 256 VGPRs each, so that the mixed-radix butterflies inside the walk over the lines do not spill
constexpr int MAX_STAGES = 16;

inline bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// the stages of one line transform: radix[0] is the innermost (first decimation-in-time) stage
struct AxisPlan {
    int32_t len;                 // transform length: n, or the Bluestein length M
    int32_t nf;                  // number of stages; 0 = power of two (the radix-2 kernel)
    uint8_t radix[MAX_STAGES];
};
// true when `len` has no prime factor above 13; fills the stage list (4s first, then 2, 3, 5, 7, 11, 13)
inline bool factor_plan(int len, AxisPlan &pl) {
    pl = AxisPlan{};
    pl.len = len;
    if (len < 1) return false;
    if (is_pow2(len)) return true;
    int rest = len, nf = 0;
    while (rest % 4 == 0) {
        if (nf == MAX_STAGES) return false;
        pl.radix[nf++] = 4;
        rest /= 4;
    }
    for (int p : {2, 3, 5, 7, 11, 13})
        while (rest % p == 0) {
            if (nf == MAX_STAGES) return false;
            pl.radix[nf++] = (uint8_t)p;
            rest /= p;
        }
    if (rest != 1) return false;
    pl.nf = nf;
    return true;
}
// rough cost of one pass of a stage, per point: the LDS round trip (1) + the butterfly's arithmetic (the 7 / 11 / 13 ones
// are plain O(r^2) sums)
inline double stage_cost(int r) {
    switch (r) {
        case 2: return 2.0;
        case 4: return 2.2;
        case 3: return 2.3;
        case 5: return 3.0;
        case 7: return 9.0;
        case 11: return 15.0;
        default: return 18.0;
    }
}
inline double plan_cost(const AxisPlan &pl) {
    if (!pl.nf) {   // power of two: the fused radix-2 kernel, two stages per LDS pass
        int lg = 0;
        while ((1 << lg) < pl.len) ++lg;
        return pl.len * (1.1 * lg);
    }
    double c = 0.0;
    for (int s = 0; s < pl.nf; ++s) c += stage_cost(pl.radix[s]);
    return pl.len * c;
}
// Bluestein length of an n-point line: 0 when n itself is smooth (no chirp needed), else the CHEAPEST smooth M in
// [2n - 1, 2n - 1 + n/4] (not simply the smallest: 2084 -> 4320 = 2^5 3^3 5 rather than 4200 = 2^3 3 5^2 7, whose radix-7
// stage costs more than the 3 % of extra points; 3122 -> 6400 = 2^8 5^2 rather than 6250 = 2 5^5); -1 when none fits a line
inline int bluestein_m(int n) {
    AxisPlan pl;
    if (factor_plan(n, pl)) return 0;
    // a length that fits the LDS if there is one, else one for a line in the workspace
    for (int cap : {MAX_LINE, MAX_LONG_LINE}) {
        int best = -1;
        double best_cost = 0.0;
        const int hi = std::min(cap, 2 * n - 1 + n / 4);
        for (int m = 2 * n - 1; m <= hi; ++m) {
            if (!factor_plan(m, pl)) continue;
            const double c = plan_cost(pl);
            if (best < 0 || c < best_cost) {
                best = m;
                best_cost = c;
            }
        }
        if (best >= 0) return best;
    }
    return -1;
}
inline bool line_supported(int n) { return n >= 2 && n <= MAX_CROP_SIDE && bluestein_m(n) >= 0; }
inline int64_t align16(int64_t v) { return (v + 15) & ~int64_t(15); }

// Workspace carve-up, identical on host (sizes) and device (pointers).
struct Layout {
    int64_t tw0, tw1, up0, up1, spectra, amps, rowmax, d1, ccup, peak, total;
    int64_t twm0, twm1, chirp0, chirp1, cspec0, cspec1;   // Bluestein tables per axis (unused for a smooth length)
    int64_t perm0, perm1;        // position of frequency k in a mixed-radix line (int32 per point; unused otherwise)
    int n0, n1, n1h, region, up;
    int sp;          // row pitch of the half spectra in elements: n1h rounded up to whole 128-byte lines (8 complex doubles)
    int m0, m1;      // Bluestein FFT length per axis, 0 = the axis length is smooth (transformed directly)
    int long0, long1;            // the axis' line (m or n points) does not fit the LDS: transformed in a scratch line of the workspace
    int n_slots;                 // scratch lines (0 when no axis is long)
    int64_t scratch, slot_bytes; // the scratch lines, LONG_SLOTS x the longer long line
    AxisPlan ax0, ax1;           // stages of the axis' transform (of length n, or of the Bluestein length m)
    int64_t per_pair_spec;
};

Layout make_layout(int n_pairs, int n0, int n1, int up) {
    Layout L{};
    L.n0 = n0;
    L.n1 = n1;
    L.n1h = n1 / 2 + 1;
    L.sp = (L.n1h + 7) & ~7;
    L.up = up;
    L.region = (int)((up * 3 + 1) / 2);   // ceil(up * 1.5) for integer up (skimage :233)
    int64_t off = 0;
    L.tw0 = off;
    off += align16((int64_t)n0 * 16);
    L.tw1 = off;
    off += align16((int64_t)n1 * 16);
    L.up0 = off;
    off += align16((int64_t)n0 * up * 16);
    L.up1 = off;
    off += align16((int64_t)n1 * up * 16);
    L.m0 = std::max(0, bluestein_m(n0));      // (callers have checked line_supported)
    L.m1 = std::max(0, bluestein_m(n1));
    factor_plan(L.m0 ? L.m0 : n0, L.ax0);
    factor_plan(L.m1 ? L.m1 : n1, L.ax1);
    L.twm0 = off;
    off += (int64_t)L.m0 * 16;
    L.twm1 = off;
    off += (int64_t)L.m1 * 16;
    L.chirp0 = off;
    off += L.m0 ? align16((int64_t)n0 * 16) : 0;
    L.chirp1 = off;
    off += L.m1 ? align16((int64_t)n1 * 16) : 0;
    L.cspec0 = off;
    off += (int64_t)L.m0 * 16;
    L.cspec1 = off;
    off += (int64_t)L.m1 * 16;
    L.perm0 = off;
    off += (!L.m0 && L.ax0.nf) ? align16((int64_t)n0 * 4) : 0;
    L.perm1 = off;
    off += (!L.m1 && L.ax1.nf) ? align16((int64_t)n1 * 4) : 0;
    L.per_pair_spec = (int64_t)n0 * L.sp * 16;
    L.long0 = (L.m0 ? L.m0 : n0) > MAX_LINE;
    L.long1 = (L.m1 ? L.m1 : n1) > MAX_LINE;
    L.n_slots = (L.long0 || L.long1) ? LONG_SLOTS : 0;
    L.slot_bytes = (int64_t)std::max(L.long0 ? (L.m0 ? L.m0 : n0) : 0, L.long1 ? (L.m1 ? L.m1 : n1) : 0) * 16;
    off = (off + 127) & ~int64_t(127);
    L.scratch = off;
    off += L.n_slots * L.slot_bytes;
    off = (off + 127) & ~int64_t(127);      // rows of the spectra start on 128-byte lines (of a 128-byte-aligned workspace)
    L.spectra = off;
    off += 2 * L.per_pair_spec * n_pairs;
    L.amps = off;
    off += align16((int64_t)n_pairs * L.n1h * 16);            // (|F|^2, |G|^2) per column
    L.rowmax = off;
    off += align16((int64_t)n_pairs * ((n0 + 1) / 2) * 16);   // (|cc| value, flat index) per row pair
    L.d1 = off;
    off += align16((int64_t)n_pairs * n0 * L.region * 16);
    L.ccup = off;
    off += align16((int64_t)n_pairs * L.region * L.region * 16);   // upsampled correlation, R x R complex
    L.peak = off;
    off += align16((int64_t)n_pairs * 16);                    // 4 ints per pair
    L.total = off;
    return L;
}

struct RegParams {
    const void *const *tile_ptrs;
    const void *tile_base;
    int64_t tile_stride;
    int32_t tile_pitch;
    const uint32_t *minmax;
    const sq_pair *pairs;
    sq_pair_result *results;
    char *ws;
    Layout L;
    int32_t n_pairs, normalization, tc, rl_fwd, rl_inv;   // tc: columns per block (K2); rl_*: lines per block (K1, K3)
    int32_t n_tiles, tile_h, tile_w;
    int32_t share;         // K2: blocks that share 128-byte lines of the spectra (see columns_kernel)
};

// everything a line transform along one axis needs (see lines_fft)
struct Axis {
    int n;                 // data length
    int m;                 // 0: transformed directly; else the Bluestein length
    int ld;                // line pitch in LDS = the transform length (m ? m : n)
    AxisPlan pl;           // stages of the transform of length ld
    const cplx *tw;        // exp(-2 pi i k / n), k < n
    const cplx *twm;       // exp(-2 pi i k / m)                    (Bluestein)
    const cplx *w;         // chirp w[j] = exp(-i pi j^2 / n), j < n  (Bluestein)
    const cplx *spec;      // FFT_m of conj(w) laid out circularly, in the order the forward m-point transform leaves it
    const int *perm;       // mixed-radix direct transform: frequency k sits at position perm[k]; else NULL (identity)
    int rev_bits;          // > 0: a power of two transformed directly -- the kernels that fill a line put element j at the
                           // bit-reversed position (put_pos) and lines_fft<.., PRE = true> skips its reversal pass; else 0
};

__device__ __forceinline__ Axis axis_of(const RegParams &P, int axis) {
    const Layout &L = P.L;
    Axis X;
    X.n = axis ? L.n1 : L.n0;
    X.m = axis ? L.m1 : L.m0;
    X.ld = X.m ? X.m : X.n;
    X.pl = axis ? L.ax1 : L.ax0;
    X.tw = reinterpret_cast<const cplx *>(P.ws + (axis ? L.tw1 : L.tw0));
    X.twm = reinterpret_cast<const cplx *>(P.ws + (axis ? L.twm1 : L.twm0));
    X.w = reinterpret_cast<const cplx *>(P.ws + (axis ? L.chirp1 : L.chirp0));
    X.spec = reinterpret_cast<const cplx *>(P.ws + (axis ? L.cspec1 : L.cspec0));
    X.perm = (!X.m && X.pl.nf) ? reinterpret_cast<const int *>(P.ws + (axis ? L.perm1 : L.perm0)) : nullptr;
    X.rev_bits = (!X.m && !X.pl.nf && X.n >= 2) ? 31 - __clz(X.n) : 0;
    return X;
}
// where frequency k of a transformed line sits (and where it has to be put before the second transform)
__device__ __forceinline__ int pos_of(const Axis &X, int k) { return X.perm ? X.perm[k] : k; }
// where a kernel that fills a line for lines_fft<.., PRE = true> puts the element that belongs at position p: the
// power-of-two transform starts with a bit-reversal pass over the whole line (an LDS round trip and a barrier), which a
// store to the reversed position in the first place makes unnecessary
__device__ __forceinline__ int put_pos(const Axis &X, int p) { return X.rev_bits ? (int)(__brev((unsigned)p) >> (32 - X.rev_bits)) : p; }

// A pair whose tile index or crop origin would read outside its tile never touches memory: its crops
// are taken as zero and its result carries coarse = INT32_MIN (the pair table lives in device memory,
// so the host cannot validate it before the launch).
__device__ __forceinline__ bool pair_ok(const RegParams &P, const sq_pair &pr) {
    const Layout &L = P.L;
    return pr.ref_tile >= 0 && pr.ref_tile < P.n_tiles && pr.mov_tile >= 0 && pr.mov_tile < P.n_tiles &&
           pr.ref_y0 >= 0 && pr.ref_x0 >= 0 && pr.mov_y0 >= 0 && pr.mov_x0 >= 0 &&
           pr.ref_y0 + L.n0 <= P.tile_h && pr.mov_y0 + L.n0 <= P.tile_h && pr.ref_x0 + L.n1 <= P.tile_w &&
           pr.mov_x0 + L.n1 <= P.tile_w;
}

// ---------------------------------------------------------------------------------------------
// line FFT in LDS
// ---------------------------------------------------------------------------------------------
// Power of two: in-place radix-2 decimation in time, tw[k] = exp(-2 pi i k / n), k < n/2, with two
// stages fused per pass: the four points {i, i+h, i+2h, i+3h} are closed under stages s and s+1, so a
// thread carries them through both in registers -- the arithmetic (and so every bit of the result)
// is that of the plain radix-2 schedule, with half the LDS passes and barriers.  All `nlines` lines
// of the block (contiguous, n points each) go through every pass together: one barrier per pass for
// the whole batch instead of one per line.
// INV conjugates the twiddles (no 1/n scaling anywhere: only argmax and ratios are used).
template <bool INV>
__device__ __forceinline__ cplx rot90(cplx a) {   // a * (-i) forward, a * (+i) inverse
    return INV ? cplx{-a.im, a.re} : cplx{a.im, -a.re};
}
template <bool INV>
__device__ __forceinline__ cplx twiddle(const cplx *__restrict__ tw, int idx) {
    cplx w = tw[idx];
    if (INV) w.im = -w.im;
    return w;
}
// (The half twiddle table copied into LDS per block -- ds_read instead of L1-cached global loads -- was built and
// measured in round 3: 1 984 pairs of 1024 x 256 in 33.3 instead of 30.0 ms, 42-44 k instead of 46-50 k pairs/s in
// 30-pair batches.  The table is L1-resident; the LDS copy only costs occupancy and LDS bandwidth.  Not kept.)
template <bool INV, bool PRE = false, bool PREFETCH = true>
__device__ void lines_fft_pow2(cplx *base, int n, int nlines, const cplx *__restrict__ tw, int tid, int nt) {
    const int logn = 31 - __clz(n);
    if (!PRE) {      // (PRE: the caller stored the line bit-reversed already, put_pos)
        for (int e = tid; e < nlines * n; e += nt) {
            const int i = e & (n - 1);
            const int j = logn ? (int)(__brev((unsigned)i) >> (32 - logn)) : 0;
            if (i < j) {
                cplx *x = base + (e - i);
                const cplx a = x[i];
                x[i] = x[j];
                x[j] = a;
            }
        }
        __syncthreads();
    }
    int s = 1;
    if (logn & 1) {   // odd number of stages: the first one alone (twiddle 1)
        for (int e = tid; e < nlines * (n >> 1); e += nt) {
            cplx *x = base + 2 * (int64_t)e;
            const cplx t = cmul(twiddle<INV>(tw, 0), x[1]);
            const cplx u = x[0];
            x[0] = cadd(u, t);
            x[1] = csub(u, t);
        }
        __syncthreads();
        s = 2;
    }
    const int quads = n >> 2, total = nlines * quads;
    // A pass is short -- one or two butterflies per thread when a block holds a line or two -- and the table read of its
    // twiddle sits right on its critical path (an L1 / L2 round trip before the first product).  The twiddles of the NEXT
    // pass depend on nothing but the butterfly's index, so this pass issues their reads before it touches its own data:
    // they arrive while the pass runs and the barrier is waited for.  (The first PF butterflies of a thread; a thread
    // with more of them reads the others' twiddles when it gets there, as before.  PREFETCH = false: the instantiations
    // inside the general (mixed-radix) kernels, which sit at their register limit -- 8 more doubles cost them a wave:
    // 312 x 3122 crops 7.2 -> 6.3 k pairs/s.)
    constexpr int PF = PREFETCH ? 2 : 0;
    cplx wcur[PF ? PF : 1], wnxt[PF ? PF : 1];
    bool primed = false;
    auto twiddle_of = [&](int s_, int e) {      // w2 of butterfly e in the pass that fuses stages s_ and s_ + 1
        const int q = e & (quads - 1), k = q & ((1 << (s_ - 1)) - 1);
        return twiddle<INV>(tw, k * (n >> (s_ + 1)));
    };
    for (; s < logn; s += 2) {
        const int half = 1 << (s - 1);
        const int ts2 = n >> (s + 1);      // (ts1 = n >> s = 2 ts2 is the stride of the stage-s twiddle)
        const bool more = s + 2 < logn;
        if (more) {
#pragma unroll
            for (int i = 0; i < PF; ++i)
                if (tid + i * nt < total) wnxt[i] = twiddle_of(s + 2, tid + i * nt);
        }
        auto butterfly4 = [&](int e, cplx w2pre, bool have) {
            const int l = e >> (logn - 2), q = e & (quads - 1);
            const int k = q & (half - 1);
            cplx *x = base + (int64_t)l * n + (((q >> (s - 1)) << (s + 1)) + k);
            // one table read per butterfly: (k + half) ts2 = k ts2 + n / 4 and k ts1 = 2 k ts2, so w3 = -i w2 (+i inverse)
            // and w1 = w2^2 -- the reads go through the L1 at 16 cycles per wave (three of them were 2.3e7 of the column
            // kernel's 2.7e7 vector memory reads per 992-pair batch), the square is four VALU operations
            const cplx x0 = x[0], x1 = x[half], x2 = x[2 * half], x3 = x[3 * half];
            cplx a0, a1, a2, a3, u2, u3;
            if (half == 1) {      // the first pass of an even number of stages: k = 0, every twiddle is 1 or -i -- no table read, no product
                a0 = cadd(x0, x1), a1 = csub(x0, x1), a2 = cadd(x2, x3), a3 = csub(x2, x3);
                u2 = a2;
                u3 = rot90<INV>(a3);
            } else {
                const cplx w2 = have ? w2pre : twiddle<INV>(tw, k * ts2), w3 = rot90<INV>(w2), w1 = cmul(w2, w2);
                // stage s: (x0, x1) and (x2, x3), both with w1
                const cplx t1 = cmul(w1, x1), t3 = cmul(w1, x3);
                a0 = cadd(x0, t1), a1 = csub(x0, t1), a2 = cadd(x2, t3), a3 = csub(x2, t3);
                // stage s + 1: (a0, a2) with w2, (a1, a3) with w3
                u2 = cmul(w2, a2), u3 = cmul(w3, a3);
            }
            x[0] = cadd(a0, u2);
            x[2 * half] = csub(a0, u2);
            x[half] = cadd(a1, u3);
            x[3 * half] = csub(a1, u3);
        };
        int e = tid;
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (e < total) {
                butterfly4(e, wcur[i], primed);
                e += nt;
            }
        for (; e < total; e += nt) butterfly4(e, cplx{0.0, 0.0}, false);
        __syncthreads();
        if (more) {
#pragma unroll
            for (int i = 0; i < PF; ++i) wcur[i] = wnxt[i];
            primed = true;
        }
    }
}

// Smooth lengths: mixed-radix Cooley-Tukey in place, radices from the AxisPlan (radix[0] = the innermost stage).
// Two forms of the same factorisation, so that no digit-reversal pass is ever needed:
//   DIF (decimation in frequency): natural-order input  -> output in "plan order" (frequency k at position perm[k]);
//   DIT (decimation in time):      plan-order input     -> natural-order output.
// A forward transform runs as DIF, the transform that follows it (pointwise work happens in plan order, positions
// looked up through perm) as DIT.  With m_s = radix[0] ... radix[s], a block of m_s points holds radix[s] sub-blocks of
// m_(s-1) points; butterfly (block, k) takes the points {k + q m_(s-1)}, q < r:
//   DIT: a_q = x_q W_ms^(q k),  y_j = sum_q a_q W_r^(q j);      DIF: y_j = (sum_q x_q W_r^(q j)) W_ms^(j k).
// Twiddles come from the full-circle table tw[t] = exp(-2 pi i t / N): W_ms^e = tw[e N / m_s] (e < m_s), W_r^e = tw[e N / r].
// float64 throughout; INV conjugates every twiddle (no 1/N anywhere, like the power-of-two kernel).
template <int R, bool INV>
__device__ __forceinline__ void small_dft(cplx (&a)[R], const cplx *__restrict__ tw, int N) {
    if constexpr (R == 2) {
        const cplx u = a[0], v = a[1];
        a[0] = cadd(u, v);
        a[1] = csub(u, v);
    } else if constexpr (R == 4) {
        const cplx t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]), t2 = cadd(a[1], a[3]), t3 = rot90<INV>(csub(a[1], a[3]));
        a[0] = cadd(t0, t2);
        a[2] = csub(t0, t2);
        a[1] = cadd(t1, t3);
        a[3] = csub(t1, t3);
    } else if constexpr (R == 3) {
        const double S3 = 0.86602540378443864676;
        const cplx t1 = cadd(a[1], a[2]);
        const cplx t2 = {a[0].re - 0.5 * t1.re, a[0].im - 0.5 * t1.im};
        const cplx d = csub(a[1], a[2]);
        const cplx t3 = rot90<INV>(cplx{S3 * d.re, S3 * d.im});
        a[0] = cadd(a[0], t1);
        a[1] = cadd(t2, t3);
        a[2] = csub(t2, t3);
    } else if constexpr (R == 5) {
        const double C1 = 0.30901699437494742410, C2 = -0.80901699437494742410;
        const double S1 = 0.95105651629515357212, S2 = 0.58778525229247312917;
        const cplx t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]), t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
        const cplx m1 = {a[0].re + C1 * t1.re + C2 * t2.re, a[0].im + C1 * t1.im + C2 * t2.im};
        const cplx m2 = {a[0].re + C2 * t1.re + C1 * t2.re, a[0].im + C2 * t1.im + C1 * t2.im};
        const cplx n1 = rot90<INV>(cplx{S1 * t3.re + S2 * t4.re, S1 * t3.im + S2 * t4.im});
        const cplx n2 = rot90<INV>(cplx{S2 * t3.re - S1 * t4.re, S2 * t3.im - S1 * t4.im});
        a[0] = cadd(a[0], cadd(t1, t2));
        a[1] = cadd(m1, n1);
        a[4] = csub(m1, n1);
        a[2] = cadd(m2, n2);
        a[3] = csub(m2, n2);
    }
}
template <int R, bool INV, bool DIF>
__device__ __forceinline__ void butterfly(cplx *x, int stride, int k, int tws, int N, const cplx *__restrict__ tw) {
    cplx a[R];
#pragma unroll
    for (int q = 0; q < R; ++q) a[q] = x[q * stride];
    if (!DIF && k) {
#pragma unroll
        for (int q = 1; q < R; ++q) a[q] = cmul(twiddle<INV>(tw, q * k * tws), a[q]);
    }
    if constexpr (R <= 5) {
        small_dft<R, INV>(a, tw, N);
        if (DIF && k) {
#pragma unroll
            for (int j = 1; j < R; ++j) a[j] = cmul(twiddle<INV>(tw, j * k * tws), a[j]);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) x[j * stride] = a[j];
    } else {
        // 7, 11, 13: the plain sum with table twiddles W_R^((q j) mod R); every input is in registers, so each output
        // goes straight back to its place (an output array would double the registers: 13 complex128 = 52 VGPRs)
        const int step = N / R;
#pragma unroll 1
        for (int j = 0; j < R; ++j) {      // (not unrolled: R outputs' worth of hoisted twiddles would fill the register file)
            cplx acc = a[0];
#pragma unroll
            for (int q = 1; q < R; ++q) acc = cadd(acc, cmul(a[q], twiddle<INV>(tw, ((q * j) % R) * step)));
            if (DIF && k && j) acc = cmul(twiddle<INV>(tw, j * k * tws), acc);
            x[j * stride] = acc;
        }
    }
}
// `nlines` contiguous lines of N points each; the caller has synchronised; returns synchronised
template <bool INV, bool DIF>
__device__ void lines_fft_mixed(cplx *base, int N, int nlines, const cplx *__restrict__ tw, const AxisPlan &pl, int tid, int nt) {
    int m_prev = 1, m = N;
    for (int step = 0; step < pl.nf; ++step) {
        const int s = DIF ? pl.nf - 1 - step : step;
        const int r = pl.radix[s];
        if (DIF) m_prev = m / r;
        const int nb = N / r, tws = N / (m_prev * r);
        for (int e = tid; e < nlines * nb; e += nt) {
            const int l = e / nb, t = e - l * nb;
            const int blk = t / m_prev, k = t - blk * m_prev;
            cplx *x = base + (int64_t)l * N + blk * (m_prev * r) + k;
            switch (r) {
                case 2: butterfly<2, INV, DIF>(x, m_prev, k, tws, N, tw); break;
                case 3: butterfly<3, INV, DIF>(x, m_prev, k, tws, N, tw); break;
                case 4: butterfly<4, INV, DIF>(x, m_prev, k, tws, N, tw); break;
                case 5: butterfly<5, INV, DIF>(x, m_prev, k, tws, N, tw); break;
                case 7: butterfly<7, INV, DIF>(x, m_prev, k, tws, N, tw); break;
                case 11: butterfly<11, INV, DIF>(x, m_prev, k, tws, N, tw); break;
                default: butterfly<13, INV, DIF>(x, m_prev, k, tws, N, tw); break;
            }
        }
        __syncthreads();
        if (DIF) m = m_prev;
        else m_prev *= r;
    }
}
// one transform of length pl.len: natural -> plan order (BWD = false) or plan order -> natural (BWD = true); for a
// power of two both orders are the natural one
// GEN = false: the instantiation for power-of-two transforms only (the host picks it when the axis' plan has no
// mixed-radix stage) -- the radix-13 butterfly alone holds 13 complex128 in registers, and a kernel that merely CONTAINS
// it is allocated for it: the power-of-two kernels would drop from 8+ to 2 waves per SIMD (measured: 1 984 pairs of
// 1024 x 256 in 57 instead of 29 ms)
template <bool INV, bool BWD, bool GEN, bool PRE = false>
__device__ __forceinline__ void lines_fft_plan(cplx *base, int nlines, const cplx *__restrict__ tw, const AxisPlan &pl, int tid, int nt) {
    if (GEN && pl.nf) lines_fft_mixed<INV, !BWD>(base, pl.len, nlines, tw, pl, tid, nt);
    else lines_fft_pow2<INV, PRE, !GEN>(base, pl.len, nlines, tw, tid, nt);
}

// Any other length n (a prime factor above 13): Bluestein's chirp-z form of the same DFT, in place in a line of
// M >= 2n - 1 points (M smooth): with w[j] = exp(-i pi j^2 / n),
//     X[k] = w[k] * sum_j (x[j] w[j]) * conj(w)[k - j]
// i.e. multiply by the chirp, convolve with the conjugate chirp (forward FFT_M, multiply by the chirp's precomputed
// spectrum -- stored in the order the forward transform leaves its output in, so nothing is permuted -- inverse FFT_M,
// 1/M), multiply by the chirp again.  float64 throughout; the chirp phases are reduced exactly in integers
// (j^2 mod 2n) before sincospi.  The inverse transform is conj(FFT(conj x)).  This is how pocketfft -- the
// reference's FFT -- treats lengths with large prime factors too (a 6244 x 4168 sensor gives crops 2084 = 4 * 521 and
// 3122 = 2 * 7 * 223 long).
//
// lines_fft: `nlines` lines at pitch X.ld; data in the first n points of each line.
//   BWD = false ("first" transform): natural-order input  -> frequency k at position pos_of(X, k);
//   BWD = true  ("second"):          input with frequency k at pos_of(X, k) -> natural-order output.
// pos_of is the identity except for a mixed-radix direct transform.
// PRE: the caller has filled the lines through put_pos (it matters for a directly transformed power of two only)
template <bool INV, bool BWD, bool GEN, bool PRE = false>
__device__ void lines_fft(cplx *base, const Axis &X, int nlines, int tid, int nt) {
    const int n = X.n;
    if (!X.m) {
        if (PRE && X.rev_bits) lines_fft_plan<INV, BWD, GEN, true>(base, nlines, X.tw, X.pl, tid, nt);
        else lines_fft_plan<INV, BWD, GEN, false>(base, nlines, X.tw, X.pl, tid, nt);
        return;
    }
    const int M = X.m;
    for (int e = tid; e < nlines * M; e += nt) {
        const int j = e % M;
        cplx v = {0.0, 0.0};
        if (j < n) {
            v = base[e];
            if (INV) v.im = -v.im;
            v = cmul(v, X.w[j]);
        }
        base[e] = v;
    }
    __syncthreads();
    lines_fft_plan<false, false, GEN>(base, nlines, X.twm, X.pl, tid, nt);
    for (int e = tid; e < nlines * M; e += nt) base[e] = cmul(base[e], X.spec[e % M]);
    __syncthreads();
    lines_fft_plan<true, true, GEN>(base, nlines, X.twm, X.pl, tid, nt);
    const double inv_m = 1.0 / (double)M;
    for (int e = tid; e < nlines * M; e += nt) {
        const int j = e % M;
        if (j < n) {
            cplx v = cmul(base[e], X.w[j]);
            v.re *= inv_m;
            v.im *= inv_m;
            if (INV) v.im = -v.im;
            base[e] = v;
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// K0: tables
// ---------------------------------------------------------------------------------------------
__global__ void init_tables_kernel(RegParams P) {
    const Layout &L = P.L;
    cplx *tw0 = reinterpret_cast<cplx *>(P.ws + L.tw0);
    cplx *tw1 = reinterpret_cast<cplx *>(P.ws + L.tw1);
    cplx *up0 = reinterpret_cast<cplx *>(P.ws + L.up0);
    cplx *up1 = reinterpret_cast<cplx *>(P.ws + L.up1);
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = gid; k < L.n0; k += stride) {
        double s, c;
        sincospi(-2.0 * (double)k / (double)L.n0, &s, &c);
        tw0[k] = {c, s};
    }
    for (int64_t k = gid; k < L.n1; k += stride) {
        double s, c;
        sincospi(-2.0 * (double)k / (double)L.n1, &s, &c);
        tw1[k] = {c, s};
    }
    for (int axis = 0; axis < 2; ++axis) {   // Bluestein tables of the axes that need them
        const int n = axis ? L.n1 : L.n0, M = axis ? L.m1 : L.m0;
        if (!M) continue;
        cplx *twm = reinterpret_cast<cplx *>(P.ws + (axis ? L.twm1 : L.twm0));
        cplx *w = reinterpret_cast<cplx *>(P.ws + (axis ? L.chirp1 : L.chirp0));
        for (int64_t k = gid; k < M; k += stride) {
            double sn, cs;
            sincospi(-2.0 * (double)k / (double)M, &sn, &cs);
            twm[k] = {cs, sn};
        }
        for (int64_t j = gid; j < n; j += stride) {
            double sn, cs;
            sincospi(-(double)((j * j) % (2 * (int64_t)n)) / (double)n, &sn, &cs);   // exp(-i pi j^2 / n), phase reduced exactly
            w[j] = {cs, sn};
        }
    }
    for (int axis = 0; axis < 2; ++axis) {   // plan-order positions of the mixed-radix direct transforms
        const int n = axis ? L.n1 : L.n0, M = axis ? L.m1 : L.m0;
        const AxisPlan &pl = axis ? L.ax1 : L.ax0;
        if (M || !pl.nf) continue;
        int *perm = reinterpret_cast<int *>(P.ws + (axis ? L.perm1 : L.perm0));
        for (int64_t i = gid; i < n; i += stride) {
            // i = q_nf + r_nf (q_(nf-1) + r_(nf-1) (... q_1)), the last stage's digit least significant; frequency
            // (and, for the DIT input, sample) i sits at sum_s q_s m_(s-1), m_(s-1) = radix[0] ... radix[s-2]
            int t = (int)i, pos = 0, m_prev = n;
            for (int s = pl.nf - 1; s >= 0; --s) {
                const int r = pl.radix[s];
                m_prev /= r;
                pos += (t % r) * m_prev;
                t /= r;
            }
            perm[i] = pos;
        }
    }
    const int64_t m0 = (int64_t)L.n0 * L.up, m1 = (int64_t)L.n1 * L.up;
    for (int64_t j = gid; j < m0; j += stride) {
        double s, c;
        sincospi(2.0 * (double)j / (double)m0, &s, &c);
        up0[j] = {c, s};
    }
    for (int64_t j = gid; j < m1; j += stride) {
        double s, c;
        sincospi(2.0 * (double)j / (double)m1, &s, &c);
        up1[j] = {c, s};
    }
}

// spectrum of the conjugate chirp, once per axis and launch: block 0 = axis 0, block 1 = axis 1.  Left in the order
// the forward M-point transform produces (plan order), which is the order lines_fft multiplies in.
// LONG: a chirp too long for the LDS is transformed where it ends up, in the workspace (axis given by the launch).
template <bool LONG>
__global__ __launch_bounds__(LONG ? 1024 : 256) void init_chirp_kernel(RegParams P, int only_axis) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout &L = P.L;
    const int axis = only_axis >= 0 ? only_axis : (int)blockIdx.x;
    const int n = axis ? L.n1 : L.n0, M = axis ? L.m1 : L.m0;
    if (!M || (bool)(axis ? L.long1 : L.long0) != LONG) return;
    const cplx *w = reinterpret_cast<const cplx *>(P.ws + (axis ? L.chirp1 : L.chirp0));
    const cplx *twm = reinterpret_cast<const cplx *>(P.ws + (axis ? L.twm1 : L.twm0));
    cplx *spec = reinterpret_cast<cplx *>(P.ws + (axis ? L.cspec1 : L.cspec0));
    cplx *x = LONG ? spec : reinterpret_cast<cplx *>(smem);
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int j = tid; j < M; j += nt) {   // conj(w)[m] at m and at M - m, zero between n - 1 and M - n + 1
        cplx v = {0.0, 0.0};
        if (j < n) v = cconj(w[j]);
        else if (M - j < n) v = cconj(w[M - j]);
        x[j] = v;
    }
    __syncthreads();
    lines_fft_plan<false, false, true>(x, 1, twm, axis ? L.ax1 : L.ax0, tid, nt);
    if (!LONG)
        for (int j = tid; j < M; j += nt) spec[j] = x[j];
}

// ---------------------------------------------------------------------------------------------
// per-tile min / max (normalize_image's two reductions)
// ---------------------------------------------------------------------------------------------
__global__ void minmax_init_kernel(uint32_t *mm, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        mm[2 * i] = 0xFFFFFFFFu;
        mm[2 * i + 1] = 0u;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void minmax_kernel(const void *const *tile_ptrs, const void *tile_base,
                                                     int64_t tile_stride, int tile_h, int tile_w, int pitch,
                                                     uint32_t *mm) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int VEC = 16 / sizeof(T);
    const int t = blockIdx.y;
    const T *tile = tile_ptrs ? static_cast<const T *>(tile_ptrs[t]) : static_cast<const T *>(tile_base) + t * tile_stride;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int y = wave; y < tile_h; y += nwaves) {
        const T *row = tile + (int64_t)y * pitch;
        const int mis = (int)((reinterpret_cast<uintptr_t>(row) / sizeof(T)) & (VEC - 1));
        const int head = mis ? min(tile_w, VEC - mis) : 0;
        const int nvec = (tile_w - head) / VEC;
        for (int v = lane; v < nvec; v += 64) {
            const u32x4 px = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(
                (const __attribute__((address_space(1))) char *)(row + head + v * VEC));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (sizeof(T) == 2) {
                    const uint32_t a = px[q] & 0xFFFFu, b = px[q] >> 16;
                    lo = min(lo, min(a, b));
                    hi = max(hi, max(a, b));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t a = (px[q] >> (8 * e)) & 0xFFu;
                        lo = min(lo, a);
                        hi = max(hi, a);
                    }
                }
            }
        }
        for (int x = lane; x < head; x += 64) {
            const uint32_t a = row[x];
            lo = min(lo, a);
            hi = max(hi, a);
        }
        for (int x = head + nvec * VEC + lane; x < tile_w; x += 64) {
            const uint32_t a = row[x];
            lo = min(lo, a);
            hi = max(hi, a);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, off));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
    }
    if (lane == 0) {
        atomicMin(&mm[2 * t], lo);
        atomicMax(&mm[2 * t + 1], hi);
    }
}

// ---------------------------------------------------------------------------------------------
// K1: crop + normalise + two-for-one FFT along axis 1
// ---------------------------------------------------------------------------------------------
// v / range for the normalisation: both are integers in [0, 65535] (a pixel minus the tile's minimum, the tile's
// maximum minus its minimum), and the quotient has to be numpy's float64 one bit for bit (its product with 65535 is
// truncated to an integer).  The IEEE division is a dozen instructions, two of them quarter-rate, twice per pixel pair:
// half of the forward row kernel's VALU work.  With the reciprocal of the range worked out once per thread, one
// multiply and two fused multiply-adds (Markstein's correction) give the correctly rounded quotient on this domain --
// not a theorem used on trust: sq_selftest_normalise_divide compares it with the compiler's division for ALL
// 65536 x 65535 pairs (and range = 0: NaN either way) on the device, and the GPU tests run it.
__device__ __forceinline__ double quotient_u16(double v, double range, double rinv) {
    if (range == 0.0) return v / range;      // a constant tile (0 / 0 = NaN for its every pixel); x / 0 = inf as IEEE has it
    const double q0 = v * rinv;
    return fma(fma(-q0, range, v), rinv, q0);
}

template <typename T>
__device__ __forceinline__ double normalised_px(uint32_t px, double lo, double range, double rinv) {
    // ((img - min) / (max - min) * dtype_max).astype(dtype)   (stitcher.py:615-617)
    // range < 0 (a min > max entry in the table) means "already normalised": use the pixel as is
    if (range < 0.0) return (double)px;
    const double scale = sizeof(T) == 1 ? 255.0 : 65535.0;
    const double v = (double)px - lo;                       // exact: uint - uint, never negative
    const double q = quotient_u16(v, range, rinv) * scale;  // 0/0 -> NaN -> 0 below, as the x86 cast does
    return q == q ? (double)(T)q : 0.0;
}
template <typename T>
__device__ __forceinline__ double normalised(const T *tile, int64_t idx, double lo, double range, double rinv) {
    return normalised_px<T>((uint32_t)tile[idx], lo, range, rinv);
}
struct __attribute__((packed)) PixVec16 {   // 16 bytes of pixels at any alignment
    uint32_t w[4];
};

// One block = P.rl consecutive rows of one pair (as many as fit 64 KB of LDS, at most 8), sent through
// the line FFT together: one barrier per pass for the batch, and eight times fewer, fuller blocks
// than one row per block.
// x: the block's lines, [rl][ld] -- LDS, or one scratch line of the workspace (a long axis 1)
template <typename T, bool GEN>
__device__ __forceinline__ void rows_forward_block(const RegParams &P, cplx *x, const int pair, const int r0, const int rl) {
    const Layout &L = P.L;
    const int n1 = L.n1, n1h = L.n1h, sp = L.sp;
    const Axis X = axis_of(P, 1);
    const int ld = X.ld;                           // line pitch: the Bluestein length when n1 needs one
    const int nrow = min(rl, L.n0 - r0);
    const sq_pair pr = P.pairs[pair];
    const int tid = threadIdx.x, nt = blockDim.x;
    cplx *A = reinterpret_cast<cplx *>(P.ws + L.spectra + (int64_t)pair * 2 * L.per_pair_spec) + (int64_t)r0 * sp;
    cplx *B = A + (int64_t)L.n0 * sp;
    if (!pair_ok(P, pr)) {   // uniform per block
        for (int e = tid; e < nrow * sp; e += nt) A[e] = B[e] = {0.0, 0.0};
        return;
    }
    const T *ref = P.tile_ptrs ? static_cast<const T *>(P.tile_ptrs[pr.ref_tile])
                               : static_cast<const T *>(P.tile_base) + pr.ref_tile * P.tile_stride;
    const T *mov = P.tile_ptrs ? static_cast<const T *>(P.tile_ptrs[pr.mov_tile])
                               : static_cast<const T *>(P.tile_base) + pr.mov_tile * P.tile_stride;
    const double rlo = P.minmax[2 * pr.ref_tile], rrange = (double)P.minmax[2 * pr.ref_tile + 1] - rlo;
    const double mlo = P.minmax[2 * pr.mov_tile], mrange = (double)P.minmax[2 * pr.mov_tile + 1] - mlo;
    const double rrinv = 1.0 / rrange, mrinv = 1.0 / mrange;      // once per thread (quotient_u16)
    // uint16 tiles: 8 pixels (16 bytes, any alignment) per lane and load instead of one -- a crop row is a run of
    // consecutive pixels; the pixels after the last whole group of 8, and uint8 tiles, one at a time
    const int g8 = sizeof(T) == 2 ? n1 >> 3 : 0;
    if (g8) {
        Walk wg(tid, nt, g8);
        for (int e = tid; e < nrow * g8; e += nt, wg.next()) {
            const int l = wg.l, j0 = wg.j << 3;
            const int64_t rbase = (int64_t)(pr.ref_y0 + r0 + l) * P.tile_pitch + pr.ref_x0 + j0;
            const int64_t mbase = (int64_t)(pr.mov_y0 + r0 + l) * P.tile_pitch + pr.mov_x0 + j0;
            const PixVec16 rv = *reinterpret_cast<const PixVec16 *>(ref + rbase), mv = *reinterpret_cast<const PixVec16 *>(mov + mbase);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t rp = (rv.w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu, mp = (mv.w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
                x[(int64_t)l * ld + put_pos(X, j0 + k)] = {normalised_px<T>(rp, rlo, rrange, rrinv), normalised_px<T>(mp, mlo, mrange, mrinv)};
            }
        }
    }
    const int rest0 = g8 << 3, nrest = n1 - rest0;
    if (nrest) {
        Walk wi(tid, nt, nrest);
        for (int e = tid; e < nrow * nrest; e += nt, wi.next()) {
            const int l = wi.l, j = rest0 + wi.j;
            const int64_t rbase = (int64_t)(pr.ref_y0 + r0 + l) * P.tile_pitch + pr.ref_x0;
            const int64_t mbase = (int64_t)(pr.mov_y0 + r0 + l) * P.tile_pitch + pr.mov_x0;
            x[(int64_t)l * ld + put_pos(X, j)] = {normalised<T>(ref, rbase + j, rlo, rrange, rrinv), normalised<T>(mov, mbase + j, mlo, mrange, mrinv)};
        }
    }
    __syncthreads();
    lines_fft<false, false, GEN, true>(x, X, nrow, tid, nt);
    Walk wo(tid, nt, n1h);
    for (int e = tid; e < nrow * n1h; e += nt, wo.next()) {
        const int l = wo.l, k = wo.j;
        const cplx *xl = x + (int64_t)l * ld;
        const cplx zk = xl[pos_of(X, k)], zc = cconj(xl[pos_of(X, k ? n1 - k : 0)]);
        // A = (Z[k] + conj Z[-k]) / 2 ;  B = (Z[k] - conj Z[-k]) / (2i)
        const int64_t at = (int64_t)l * sp + k;
        A[at] = {0.5 * (zk.re + zc.re), 0.5 * (zk.im + zc.im)};
        const cplx d = {zk.re - zc.re, zk.im - zc.im};
        B[at] = {0.5 * d.im, -0.5 * d.re};
        // A constant tile normalises to zeros (0/0 -> NaN -> 0, stitcher.py:613-617) and its spectrum is exactly
        // zero in the reference; packed with the other image, the separation above would leave that image's
        // rounding noise (1e-16 of its magnitude) in it, and with nothing else in the cross-power spectrum the
        // noise would pick the peak.  (Golden case reg_blank_centre: the reference lands on index 0.)
        if (rrange == 0.0) A[at] = {0.0, 0.0};
        if (mrange == 0.0) B[at] = {0.0, 0.0};
    }
}
// the block's scratch line of the workspace (long lines)
__device__ __forceinline__ cplx *scratch_line(const RegParams &P) {
    return reinterpret_cast<cplx *>(P.ws + P.L.scratch + (int64_t)blockIdx.x * P.L.slot_bytes);
}
template <typename T, bool GEN, bool LONG = false>
__global__ __launch_bounds__(LONG ? LONG_THREADS : 1024) void rows_forward_kernel(RegParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (!LONG) {
        rows_forward_block<T, GEN>(P, reinterpret_cast<cplx *>(smem), blockIdx.y, blockIdx.x * P.rl_fwd, P.rl_fwd);
    } else {      // a grid of scratch lines walks over (pair, row)
        cplx *x = scratch_line(P);
        const int64_t total = (int64_t)P.n_pairs * P.L.n0;
        for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
            rows_forward_block<T, GEN>(P, x, (int)(w / P.L.n0), (int)(w % P.L.n0), 1);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2: columns -- FFT both spectra along axis 0, cross-power product, inverse along axis 0
// ---------------------------------------------------------------------------------------------
#ifndef SQ_COL_THREADS
#define SQ_COL_THREADS 512
#endif
// Which columns a block of the column kernels takes.  A block's pieces of a spectrum row are tc * 16 bytes, so 8 / tc
// neighbouring blocks read and write the same 128-byte lines.  Workgroups go to the 8 XCDs (each with its own L2)
// round-robin by their linear index: the blocks that share lines are made the ones an XCD receives back to back --
// linear index l, l + 8, ... -- so that the second to last find the line in that L2 instead of each fetching it from
// HBM again (FETCH_SIZE of the column kernel was 2.7 x the spectra with neighbours on different XCDs).  share = 8 / tc
// when that is 2, 4 or 8, else 1; gridDim.x is a multiple of it (blocks past the last column leave at once).
__device__ __forceinline__ void column_block(int share, int &pair, int &cb) {
    pair = blockIdx.y;
    cb = blockIdx.x;
    if (share > 1) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, span = 8 * share, total = gridDim.x * gridDim.y;
        const int base = lin / span * span;
        if (base + span <= total) {
            const int in = lin - base;
            const int v = base + (in & 7) * share + (in >> 3);
            pair = v / (int)gridDim.x;
            cb = v - pair * (int)gridDim.x;
        }
    }
}

template <bool GEN>
__global__ __launch_bounds__(SQ_COL_THREADS) void columns_kernel(RegParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Layout &L = P.L;
    const int n0 = L.n0, n1h = L.n1h, sp = L.sp, tc = P.tc;
    cplx *f = reinterpret_cast<cplx *>(smem);   // [tc][n0]
    cplx *g = f + (int64_t)tc * n0;             // [tc][n0]
    __shared__ double red[2][SQ_COL_THREADS / 64];
    int pair, cb;
    column_block(P.share, pair, cb);
    const int c0 = cb * tc;
    if (c0 >= n1h) return;
    const int ncol = min(tc, n1h - c0);
    const int tid = threadIdx.x, nt = blockDim.x;
    cplx *A = reinterpret_cast<cplx *>(P.ws + L.spectra + (int64_t)pair * 2 * L.per_pair_spec);
    cplx *B = A + (int64_t)n0 * sp;
    const Axis X = axis_of(P, 0);   // a directly transformed axis here (power of two or smooth): ld == n0
    Walk wl(tid, nt, ncol);
    for (int i = tid; i < n0 * ncol; i += nt, wl.next()) {
        const int r = wl.l, c = wl.j, at = put_pos(X, r);
        f[(int64_t)c * n0 + at] = A[(int64_t)r * sp + c0 + c];
        g[(int64_t)c * n0 + at] = B[(int64_t)r * sp + c0 + c];
    }
    __syncthreads();
    if (ncol == tc) {   // f and g are contiguous: one batch of 2 tc lines
        lines_fft<false, false, GEN, true>(f, X, 2 * tc, tid, nt);
    } else {
        lines_fft<false, false, GEN, true>(f, X, ncol, tid, nt);
        lines_fft<false, false, GEN, true>(g, X, ncol, tid, nt);
    }
    const double eps100 = 100.0 * 2.220446049250313e-16;
    double *amps = reinterpret_cast<double *>(P.ws + L.amps) + ((int64_t)pair * n1h + c0) * 2;
    for (int c = 0; c < ncol; ++c) {
        double sf = 0.0, sg = 0.0;
        for (int r = tid; r < n0; r += nt) {
            const int at = pos_of(X, r);   // frequency r of the column (plan order until the inverse transform)
            const cplx F = f[(int64_t)c * n0 + at], G = g[(int64_t)c * n0 + at];
            sf += F.re * F.re + F.im * F.im;
            sg += G.re * G.re + G.im * G.im;
            cplx pr = cmul(F, cconj(G));                              // skimage :211
            if (P.normalization == SQ_NORM_PHASE) {
                // image_product /= max(|.|, 100 eps); numpy divides complex by real as x * (1/c)
                const double scl = 1.0 / fmax(hypot(pr.re, pr.im), eps100);
                pr.re *= scl;
                pr.im *= scl;
            }
            f[(int64_t)c * n0 + at] = pr;
        }
        for (int off = 32; off > 0; off >>= 1) {
            sf += __shfl_xor(sf, off);
            sg += __shfl_xor(sg, off);
        }
        if ((tid & 63) == 0) {
            red[0][tid >> 6] = sf;
            red[1][tid >> 6] = sg;
        }
        __syncthreads();
        if (tid == 0) {
            double a = 0.0, b = 0.0;
            for (int w = 0; w < (nt >> 6); ++w) {
                a += red[0][w];
                b += red[1][w];
            }
            amps[2 * c] = a;
            amps[2 * c + 1] = b;
        }
        __syncthreads();
    }
    // the product is what the upsampled DFT reads (skimage :239): keep it in B
    Walk wb(tid, nt, ncol);
    for (int i = tid; i < n0 * ncol; i += nt, wb.next()) {
        const int r = wb.l, c = wb.j;
        B[(int64_t)r * sp + c0 + c] = f[(int64_t)c * n0 + pos_of(X, r)];
    }
    __syncthreads();
    lines_fft<true, true, GEN>(f, X, ncol, tid, nt);
    Walk wa(tid, nt, ncol);
    for (int i = tid; i < n0 * ncol; i += nt, wa.next()) {
        const int r = wa.l, c = wa.j;
        A[(int64_t)r * sp + c0 + c] = f[(int64_t)c * n0 + r];
    }
}

// The same with ONE column per block, its line (n0 points, or the Bluestein line of m0) alone in LDS -- for an axis-0
// length that needs Bluestein, and for direct lengths too long for two of them to share the LDS (n0 > 4608).  The
// column spectra F and G go back to the workspace between the three transforms instead of staying in LDS -- every
// thread re-reads exactly the elements it wrote.
template <bool GEN>
__device__ __forceinline__ void columns_single_block(const RegParams &P, cplx *x, double (&red)[2][1024 / 64], const int pair, const int c) {
    const Layout &L = P.L;
    const int n0 = L.n0, n1h = L.n1h, sp = L.sp;
    const int tid = threadIdx.x, nt = blockDim.x;
    cplx *A = reinterpret_cast<cplx *>(P.ws + L.spectra + (int64_t)pair * 2 * L.per_pair_spec) + c;
    cplx *B = A + (int64_t)n0 * sp;
    const Axis X = axis_of(P, 0);
    for (int r = tid; r < n0; r += nt) x[r] = A[(int64_t)r * sp];
    __syncthreads();
    lines_fft<false, false, GEN>(x, X, 1, tid, nt);
    for (int r = tid; r < n0; r += nt) A[(int64_t)r * sp] = x[pos_of(X, r)];   // F, natural order; re-read below by this very thread
    __syncthreads();
    for (int r = tid; r < n0; r += nt) x[r] = B[(int64_t)r * sp];
    __syncthreads();
    lines_fft<false, false, GEN>(x, X, 1, tid, nt);
    const double eps100 = 100.0 * 2.220446049250313e-16;
    double sf = 0.0, sg = 0.0;
    for (int r = tid; r < n0; r += nt) {
        const int at = pos_of(X, r);
        const cplx F = A[(int64_t)r * sp], G = x[at];
        sf += F.re * F.re + F.im * F.im;
        sg += G.re * G.re + G.im * G.im;
        cplx pr = cmul(F, cconj(G));                              // skimage :211
        if (P.normalization == SQ_NORM_PHASE) {
            const double scl = 1.0 / fmax(hypot(pr.re, pr.im), eps100);
            pr.re *= scl;
            pr.im *= scl;
        }
        B[(int64_t)r * sp] = pr;              // the product is what the upsampled DFT reads (skimage :239)
        x[at] = pr;
    }
    for (int off = 32; off > 0; off >>= 1) {
        sf += __shfl_xor(sf, off);
        sg += __shfl_xor(sg, off);
    }
    if ((tid & 63) == 0) {
        red[0][tid >> 6] = sf;
        red[1][tid >> 6] = sg;
    }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < (nt >> 6); ++w) {
            a += red[0][w];
            b += red[1][w];
        }
        double *amps = reinterpret_cast<double *>(P.ws + L.amps) + ((int64_t)pair * n1h + c) * 2;
        amps[0] = a;
        amps[1] = b;
    }
    lines_fft<true, true, GEN>(x, X, 1, tid, nt);
    for (int r = tid; r < n0; r += nt) A[(int64_t)r * sp] = x[r];
}
template <bool GEN, bool LONG = false>
__global__ __launch_bounds__(LONG ? LONG_THREADS : 1024) void columns_single_kernel(RegParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[2][1024 / 64];
    if constexpr (!LONG) {
        int pair, c;
        column_block(P.share, pair, c);
        if (c >= P.L.n1h) return;
        columns_single_block<GEN>(P, reinterpret_cast<cplx *>(smem), red, pair, c);
    } else {      // a grid of scratch lines walks over (pair, column)
        cplx *x = scratch_line(P);
        const int64_t total = (int64_t)P.n_pairs * P.L.n1h;
        for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
            columns_single_block<GEN>(P, x, red, (int)(w / P.L.n1h), (int)(w % P.L.n1h));
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K3: inverse along axis 1 (two real rows per complex FFT) + |.| + block argmax
// ---------------------------------------------------------------------------------------------
struct Best {
    double v;
    long long idx;
    int nan;
};
__device__ __forceinline__ Best better(Best a, Best b) {
    // numpy argmax: NaN is the maximum; first occurrence wins.  One predicate and three selects: returning one of
    // the two structs by an if / else chain made the compiler keep both in scratch memory and load the winner through a
    // selected address (69 scratch instructions in the row kernel's scan).
    const bool first = a.idx < b.idx;
    const bool take_a = a.nan != b.nan ? a.nan != 0 : (a.nan || a.v == b.v ? first : a.v > b.v);
    Best r;
    r.v = take_a ? a.v : b.v;
    r.idx = take_a ? a.idx : b.idx;
    r.nan = take_a ? a.nan : b.nan;
    return r;
}
__device__ __forceinline__ Best make_best(double v, long long idx) { return {v, idx, v != v ? 1 : 0}; }
__device__ Best block_best(Best b, int tid, int nt) {
    __shared__ double sv[4];
    __shared__ long long si[4];
    __shared__ int sn[4];
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.v = __shfl_xor(b.v, off);
        o.idx = __shfl_xor(b.idx, off);
        o.nan = __shfl_xor(b.nan, off);
        b = better(b, o);
    }
    if ((tid & 63) == 0) {
        sv[tid >> 6] = b.v;
        si[tid >> 6] = b.idx;
        sn[tid >> 6] = b.nan;
    }
    __syncthreads();
    Best r = {sv[0], si[0], sn[0]};
    for (int w = 1; w < (nt >> 6); ++w) r = better(r, Best{sv[w], si[w], sn[w]});
    __syncthreads();
    return r;
}

// One block = P.rl row pairs (lines) of one pair; the lines go through the FFT together, then every
// wave scans whole lines for their maximum (shuffles only, no block reduction) and writes one
// (value, index) per row pair, which K4a reduces.
__device__ __forceinline__ Best wave_best(Best b) {
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.v = __shfl_xor(b.v, off);
        o.idx = __shfl_xor(b.idx, off);
        o.nan = __shfl_xor(b.nan, off);
        b = better(b, o);
    }
    return b;
}

template <bool GEN>
__device__ __forceinline__ void rows_inverse_block(const RegParams &P, cplx *x, const int pair, const int rp0, const int rl) {
    const Layout &L = P.L;
    const int n0 = L.n0, n1 = L.n1, n1h = L.n1h, sp = L.sp;
    const int nrp = (n0 + 1) / 2;
    const Axis X = axis_of(P, 1);
    const int ld = X.ld;
    const int nline = min(rl, nrp - rp0);
    const int tid = threadIdx.x, nt = blockDim.x;
    const cplx *Q = reinterpret_cast<const cplx *>(P.ws + L.spectra + (int64_t)pair * 2 * L.per_pair_spec);
    Walk wq(tid, nt, n1);
    for (int e = tid; e < nline * n1; e += nt, wq.next()) {
        const int l = wq.l, k = wq.j;
        const int y0 = 2 * (rp0 + l), y1 = min(y0 + 1, n0 - 1);   // odd n0: the last line repeats its row
        const bool two = (y0 + 1) < n0;
        const cplx *q0 = Q + (int64_t)y0 * sp, *q1 = Q + (int64_t)y1 * sp;
        // Hermitian extension of the half spectrum of a real row: X[n1-k] = conj X[k]
        cplx a, b;
        if (k < n1h) {
            a = q0[k];
            b = q1[k];
        } else {
            a = cconj(q0[n1 - k]);
            b = cconj(q1[n1 - k]);
        }
        if (!two) b = {0.0, 0.0};
        x[(int64_t)l * ld + put_pos(X, pos_of(X, k))] = {a.re - b.im, a.im + b.re};   // a + i b, where the second transform wants frequency k
    }
    __syncthreads();
    lines_fft<true, true, GEN, true>(x, X, nline, tid, nt);
    const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    for (int l = wave; l < nline; l += nw) {
        const int y0 = 2 * (rp0 + l), y1 = min(y0 + 1, n0 - 1);
        const bool two = (y0 + 1) < n0;
        const cplx *xl = x + (int64_t)l * ld;
        Best best = {-1.0, (long long)1 << 62, 0};
        for (int k = lane; k < n1; k += 64) {
            best = better(best, make_best(fabs(xl[k].re), (long long)y0 * n1 + k));
            if (two) best = better(best, make_best(fabs(xl[k].im), (long long)y1 * n1 + k));
        }
        best = wave_best(best);
        if (lane == 0) {
            double *out = reinterpret_cast<double *>(P.ws + L.rowmax) + ((int64_t)pair * nrp + rp0 + l) * 2;
            out[0] = best.v;   // NaN stays NaN
            reinterpret_cast<long long *>(out)[1] = best.idx;
        }
    }
}
template <bool GEN, bool LONG = false>
__global__ __launch_bounds__(LONG ? LONG_THREADS : 1024) void rows_inverse_kernel(RegParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (!LONG) {
        rows_inverse_block<GEN>(P, reinterpret_cast<cplx *>(smem), blockIdx.y, blockIdx.x * P.rl_inv, P.rl_inv);
    } else {      // a grid of scratch lines walks over (pair, row pair)
        cplx *x = scratch_line(P);
        const int nrp = (P.L.n0 + 1) / 2;
        const int64_t total = (int64_t)P.n_pairs * nrp;
        for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
            rows_inverse_block<GEN>(P, x, (int)(w / nrp), (int)(w % nrp), 1);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K4a: whole-pixel peak
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void peak_kernel(RegParams P) {
    const Layout &L = P.L;
    const int pair = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int nrp = (L.n0 + 1) / 2;
    const double *rm = reinterpret_cast<const double *>(P.ws + L.rowmax) + (int64_t)pair * nrp * 2;
    // the column energies are summed by one thread in column order (a fixed order: the sums are
    // outputs); staging them through LDS first turns its chain of dependent global loads into one
    // coalesced read by the block
    __shared__ double amp_l[2 * (MAX_LINE / 2 + 1)];
    const double *amps = reinterpret_cast<const double *>(P.ws + L.amps) + (int64_t)pair * L.n1h * 2;
    const bool staged = L.n1h <= MAX_LINE / 2 + 1;      // (a long axis 1: summed straight from the workspace)
    if (staged)
        for (int i = tid; i < 2 * L.n1h; i += nt) amp_l[i] = amps[i];
    Best best = {-1.0, (long long)1 << 62, 0};
    for (int i = tid; i < nrp; i += nt) best = better(best, make_best(rm[2 * i], reinterpret_cast<const long long *>(rm)[2 * i + 1]));
    best = block_best(best, tid, nt);   // contains the barriers that publish amp_l
    if (tid == 0) {
        const int py = (int)(best.idx / L.n1), px = (int)(best.idx % L.n1);
        // shifts[shifts > fix(n/2)] -= n   (skimage :217-220)
        const int sy = py > L.n0 / 2 ? py - L.n0 : py;
        const int sx = px > L.n1 / 2 ? px - L.n1 : px;
        int *pk = reinterpret_cast<int *>(P.ws + L.peak) + 4 * pair;
        pk[0] = sy;
        pk[1] = sx;
        sq_pair_result &res = P.results[pair];
        const bool ok = pair_ok(P, P.pairs[pair]);
        res.coarse[0] = ok ? sy : INT32_MIN;
        res.coarse[1] = ok ? sx : INT32_MIN;
        res.fine[0] = res.fine[1] = 0;
        // |cc| at the whole-pixel peak, with ifftn's 1/(n0 n1); sign is not recoverable from |.|,
        // the upsampled stage overwrites this with the complex value when u > 1
        res.ccmax_re = best.v / ((double)L.n0 * (double)L.n1);
        res.ccmax_im = 0.0;
        double sa = 0.0, sb = 0.0;
        for (int k = 0; k < L.n1h; ++k) {
            // a column and its Hermitian mirror carry the same energy
            const double wgt = (k == 0 || (2 * k == L.n1)) ? 1.0 : 2.0;
            sa += wgt * (staged ? amp_l[2 * k] : amps[2 * k]);
            sb += wgt * (staged ? amp_l[2 * k + 1] : amps[2 * k + 1]);
        }
        if (L.up == 1) {   // skimage :224-228 divides by size in this branch only
            sa /= (double)L.n0 * (double)L.n1;
            sb /= (double)L.n0 * (double)L.n1;
        }
        res.src_amp = sa;
        res.tgt_amp = sb;
    }
}

// numpy.fft.fftfreq(n, d=u)[k] * (n*u): the signed integer frequency
__device__ __forceinline__ int signed_freq(int k, int n) { return k < (n - 1) / 2 + 1 ? k : k - n; }
__device__ __forceinline__ int posmod(long long a, int m) {
    int r = (int)(a % m);
    return r < 0 ? r + m : r;
}

// ---------------------------------------------------------------------------------------------
// K4b: D1[k0][b] = sum_{k1 < n1} P[k0][k1] * exp(+2 pi i (b - off1) f1[k1])
// ---------------------------------------------------------------------------------------------
// A small complex GEMM per pair: [n0 x n1] spectrum times the [n1 x R] phase matrix
// W[k1][b] = E[((b - off1) * f1[k1]) mod M], which depends on the pair (through its peak) but not on
// k0 -- so it is built once per block and reused by every row of the block's tile instead of being
// looked up per element (the first version did that, with a 64-bit modulo each, and reduced across
// the block once per output: 42 % of an all-pairs batch).
// Only the half spectrum k1 < n1h is stored; the other half is conj P[-k0][n1 - k1].  A block therefore takes ROWS
// rows 0 < k0 < n0 / 2 TOGETHER WITH their mirror rows -k0 (the two rows that are their own mirrors, 0 and -- for an
// even n0 -- n0 / 2, share the first slot): with both tiles in LDS,
//     D1[ k0][b] = sum_{k1 < n1h} P[k0][k1] W[k1][b] + conj(P[-k0][k1]) W[n1 - k1][b]
//     D1[-k0][b] = sum_{k1 < n1h} P[-k0][k1] W[k1][b] + conj(P[k0][k1]) W[n1 - k1][b]
// (second terms for 1 <= k1 <= n1 - n1h only), so every element of the product is read from HBM once -- the version
// that took rows one tile at a time read it twice (FETCH_SIZE 6.2 GB per 992-pair batch for a 2.1 GB product, at
// 3.9 TB/s: the kernel is bound by that read).
// Block = 64 (or 16) row pairs of one pair, 256 threads, thread tile = one row and its mirror x 4 outputs (or x 1),
// the k1 range in chunks of 16 (32 in the small variant) staged through LDS (row pitch padded by one element: the 16
// rows a wave reads at one k1 fall into distinct banks).  Every output is summed over k1 in ascending order by one
// thread: no cross-thread reduction, nothing depends on scheduling.
// TB = 4 (64 row pairs per block) for batches, 1 (16 row pairs per block) when there are too few pairs to fill the
// chip with the larger blocks (the bench's two centre pairs).
constexpr int UR_B = 16;
template <int TB, int UR_KC>
__global__ __launch_bounds__(256) void upsample_rows_kernel(RegParams P) {
    constexpr int ROWS = 256 / (UR_B / TB);
    const Layout &L = P.L;
    const int n0 = L.n0, n1 = L.n1, n1h = L.n1h, sp = L.sp, R = L.region, up = L.up;
    const int pair = blockIdx.y, row0 = blockIdx.x * ROWS;
    const int tid = threadIdx.x;
    const int *pk = reinterpret_cast<const int *>(P.ws + L.peak) + 4 * pair;
    // shifts = round(shifts*u)/u is the integer peak; offset = fix(R/2) - shift*u (skimage :232-238)
    const int off1 = R / 2 - pk[1] * up;
    const cplx *Pm = reinterpret_cast<const cplx *>(P.ws + L.spectra + (int64_t)pair * 2 * L.per_pair_spec) + (int64_t)n0 * sp;
    const cplx *E = reinterpret_cast<const cplx *>(P.ws + L.up1);
    const int M = n1 * up;
    // slot k0 holds the rows (k0, n0 - k0); slot 0 the self-mirrored rows (0, n0 / 2) [even n0] or row 0 alone [odd n0]
    const int n_slots = (n0 + 1) / 2, nyquist = (n0 & 1) ? -1 : n0 / 2;
    const int n_mirror = n1 - n1h;           // k1 = 1 .. n_mirror have a mirror column n1 - k1 >= n1h
    __shared__ cplx Pl[2][ROWS][UR_KC + 1];  // [0]: rows k0, [1]: rows -k0
    __shared__ cplx Wl[2][UR_KC][UR_B];      // [0]: W[k1], [1]: W[n1 - k1] (zero where k1 has no mirror)
    const int rt = tid / (UR_B / TB), bt = tid % (UR_B / TB);   // slot rt, outputs TB*bt .. of the chunk
    const bool self = row0 + rt == 0;
    for (int b0 = 0; b0 < R; b0 += UR_B) {
        cplx acc[2][TB];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) acc[i][j] = {0.0, 0.0};
        // a power-of-two n1 has n1h = n1 / 2 + 1 columns: one more than a whole number of chunks.  One or two columns left
        // over are not worth a chunk of their own (staging, two barriers, UR_KC steps of which one does anything: a ninth
        // chunk for the 129 columns of a 256-wide crop): each thread adds them from global memory after the loop
        const int left = n1h % UR_KC, kfull = left <= 2 ? n1h - left : n1h;
        for (int kc = 0; kc < kfull; kc += UR_KC) {
            for (int e = tid; e < 2 * ROWS * UR_KC; e += 256) {
                const int m = e / (ROWS * UR_KC), ik = e - m * (ROWS * UR_KC);
                const int i = ik / UR_KC, k = ik - i * UR_KC;
                const int k0 = row0 + i, k1 = kc + k;
                cplx v = {0.0, 0.0};
                const int row = m ? (k0 ? n0 - k0 : nyquist) : k0;
                if (k0 < n_slots && row >= 0 && k1 < n1h) v = Pm[(int64_t)row * sp + k1];
                Pl[m][i][k] = v;
            }
            for (int e = tid; e < 2 * UR_KC * UR_B; e += 256) {   // and the phase tiles
                const int m = e / (UR_KC * UR_B), kb = e - m * (UR_KC * UR_B);
                const int k = kb / UR_B, b = kb - k * UR_B;
                const int k1 = kc + k;
                cplx w = {0.0, 0.0};
                const bool live = m ? (k1 >= 1 && k1 <= n_mirror) : k1 < n1h;
                if (live && b0 + b < R) w = E[posmod((long long)(b0 + b - off1) * signed_freq(m ? n1 - k1 : k1, n1), M)];
                Wl[m][k][b] = w;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < UR_KC; ++k) {
                const cplx p = Pl[0][rt][k], q = Pl[1][rt][k];
                // the conjugated term comes from the mirror row: the other row of the slot, or the row itself in slot 0
                const cplx pm = cconj(self ? p : q), qm = cconj(self ? q : p);
#pragma unroll
                for (int j = 0; j < TB; ++j) {
                    const cplx w = Wl[0][k][TB * bt + j], wm = Wl[1][k][TB * bt + j];
                    acc[0][j] = cfma(pm, wm, cfma(p, w, acc[0][j]));
                    acc[1][j] = cfma(qm, wm, cfma(q, w, acc[1][j]));
                }
            }
            __syncthreads();
        }
        const int k0 = row0 + rt;
        for (int k1 = kfull; k1 < n1h; ++k1) {      // the left-over columns (same order of accumulation: ascending k1)
            if (k0 >= n_slots) break;
            const int km1 = k0 ? n0 - k0 : nyquist;
            const cplx p = Pm[(int64_t)k0 * sp + k1], q = km1 >= 0 ? Pm[(int64_t)km1 * sp + k1] : cplx{0.0, 0.0};
            const cplx pm = cconj(self ? p : q), qm = cconj(self ? q : p);
            const bool mirrored = k1 >= 1 && k1 <= n_mirror;
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const int b = b0 + TB * bt + j;
                cplx w = {0.0, 0.0}, wm = {0.0, 0.0};
                if (b < R) {
                    w = E[posmod((long long)(b - off1) * signed_freq(k1, n1), M)];
                    if (mirrored) wm = E[posmod((long long)(b - off1) * signed_freq(n1 - k1, n1), M)];
                }
                acc[0][j] = cfma(pm, wm, cfma(p, w, acc[0][j]));
                acc[1][j] = cfma(qm, wm, cfma(q, w, acc[1][j]));
            }
        }
        if (k0 < n_slots) {
            const int km = k0 ? n0 - k0 : nyquist;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i && km < 0) continue;         // odd n0: row 0 is alone in its slot
                cplx *D1 = reinterpret_cast<cplx *>(P.ws + L.d1) + ((int64_t)pair * n0 + (i ? km : k0)) * R;
#pragma unroll
                for (int j = 0; j < TB; ++j)
                    if (b0 + TB * bt + j < R) D1[b0 + TB * bt + j] = acc[i][j];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K4c: cc_up[a][b] = sum_{k0} D1[k0][b] * exp(+2 pi i (a - off0) f0[k0])   (one block per (a, pair))
// The k0 range is cut into 16 slices summed by 16 thread groups and combined in slice order, so the
// result does not depend on scheduling.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample_cols_kernel(RegParams P) {
    const Layout &L = P.L;
    const int n0 = L.n0, R = L.region, up = L.up;
    const int a = blockIdx.x, pair = blockIdx.y, tid = threadIdx.x;
    const int *pk = reinterpret_cast<const int *>(P.ws + L.peak) + 4 * pair;
    const int off0 = R / 2 - pk[0] * up;
    const cplx *D1 = reinterpret_cast<const cplx *>(P.ws + L.d1) + (int64_t)pair * n0 * R;
    const cplx *E = reinterpret_cast<const cplx *>(P.ws + L.up0);
    const int M = n0 * up;
    const long long ma = a - off0;
    __shared__ cplx part[16][16];
    cplx *out = reinterpret_cast<cplx *>(P.ws + L.ccup) + ((int64_t)pair * R + a) * R;
    for (int b0 = 0; b0 < R; b0 += 16) {   // R = 15 for u = 10: one pass
        const int b = b0 + (tid & 15), slice = tid >> 4;
        cplx acc = {0.0, 0.0};
        if (b < R) {
            const int k_lo = (int)((int64_t)n0 * slice / 16), k_hi = (int)((int64_t)n0 * (slice + 1) / 16);
            // the phase index advances by (a - off0) mod M from one k0 to the next, except where the
            // signed frequency wraps from positive to negative
            const int step = posmod(ma, M), wrap = (n0 - 1) / 2 + 1;
            int idx = 0;
            for (int k0 = k_lo; k0 < k_hi; ++k0) {
                if (k0 == k_lo || k0 == wrap) {
                    idx = posmod(ma * signed_freq(k0, n0), M);
                } else {
                    idx += step;
                    if (idx >= M) idx -= M;
                }
                acc = cadd(acc, cmul(D1[(int64_t)k0 * R + b], E[idx]));
            }
        }
        part[slice][tid & 15] = acc;
        __syncthreads();
        if (tid < 16 && b0 + tid < R) {
            cplx sum = part[0][tid];
            for (int sl = 1; sl < 16; ++sl) sum = cadd(sum, part[sl][tid]);
            out[b0 + tid] = sum;
        }
        __syncthreads();
    }
}

// K4d: argmax |cc_up| in row-major order (skimage :244), one block per pair
__global__ __launch_bounds__(256) void upsample_peak_kernel(RegParams P) {
    const Layout &L = P.L;
    const int R = L.region;
    const int pair = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const cplx *cc = reinterpret_cast<const cplx *>(P.ws + L.ccup) + (int64_t)pair * R * R;
    Best best = {-1.0, (long long)1 << 62, 0};
    for (int o = tid; o < R * R; o += nt) best = better(best, make_best(hypot(cc[o].re, cc[o].im), o));
    best = block_best(best, tid, nt);
    if (tid == 0) {
        sq_pair_result &res = P.results[pair];
        res.fine[0] = (int)(best.idx / R);
        res.fine[1] = (int)(best.idx % R);
        res.ccmax_re = cc[best.idx].re;
        res.ccmax_im = cc[best.idx].im;
    }
}

// ---------------------------------------------------------------------------------------------
// stand-alone normalize_image (the registration pipeline fuses the same arithmetic into K1)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void normalize_kernel(const void *const *tile_ptrs, const void *tile_base,
                                                        int64_t tile_stride, int tile_h, int tile_w, int pitch,
                                                        const uint32_t *minmax, T *out) {
    const int t = blockIdx.y;
    const T *tile = tile_ptrs ? static_cast<const T *>(tile_ptrs[t]) : static_cast<const T *>(tile_base) + t * tile_stride;
    const double lo = minmax[2 * t], range = (double)minmax[2 * t + 1] - lo, rinv = 1.0 / range;
    const int64_t n = (int64_t)tile_h * tile_w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / tile_w), x = (int)(i - (int64_t)y * tile_w);
        out[(int64_t)t * n + i] = (T)normalised<T>(tile, (int64_t)y * pitch + x, lo, range, rinv);
    }
}

// every (v, range) the normalisation can meet, v and range in [0, 65535] (range 0: the quotient is NaN or inf either way):
// quotient_u16 against the compiler's IEEE division, bit for bit
__global__ __launch_bounds__(256) void selftest_normalise_kernel(unsigned long long *bad) {
    const double range = (double)blockIdx.x, rinv = 1.0 / range;
    unsigned long long local = 0;
    for (int v = threadIdx.x; v < 65536; v += 256) {
        const double a = (double)v / range, b = quotient_u16((double)v, range, rinv);
        const bool same = __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b);
        local += same ? 0 : 1;
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(bad, local);
}

int check_line(int n, const char *axis) {
    if (n < 2) return fail(SQ_ERR_INVALID, "sq_register_pairs: crop %s length %d < 2", axis, n);
    if (!line_supported(n))
        return fail(SQ_ERR_UNSUPPORTED, "sq_register_pairs: crop %s length %d not supported: at most %d pixels a side", axis, n, MAX_CROP_SIDE);
    return SQ_OK;
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return SQ_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_register_pairs: cannot raise the LDS limit to %zu bytes: %s", bytes, hipGetErrorString(e));
    return SQ_OK;
}

int pick_threads(int n) { return n <= 128 ? 64 : (n <= 512 ? 128 : 256); }

}  // namespace

extern "C" int sq_tile_minmax(const void *const *tile_ptrs_dev, const void *tile_base_dev, int64_t tile_stride,
                              int32_t n_tiles, int32_t tile_h, int32_t tile_w, int32_t tile_pitch, int32_t tile_dtype,
                              uint32_t *out_minmax_dev, void *stream_) {
    if ((!tile_ptrs_dev && !tile_base_dev) || !out_minmax_dev || n_tiles < 0 || tile_h <= 0 || tile_w <= 0 ||
        tile_pitch < tile_w)
        return fail(SQ_ERR_INVALID, "sq_tile_minmax: bad arguments (n_tiles=%d %dx%d pitch %d)", n_tiles, tile_h, tile_w,
                    tile_pitch);
    if (tile_dtype != SQ_U8 && tile_dtype != SQ_U16) return fail(SQ_ERR_UNSUPPORTED, "sq_tile_minmax: dtype %d", tile_dtype);
    if (n_tiles == 0) return SQ_OK;
    if (n_tiles > 65535) return fail(SQ_ERR_UNSUPPORTED, "sq_tile_minmax: more than 65535 tiles per call");
    hipStream_t s = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(minmax_init_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, s, out_minmax_dev, n_tiles);
    const int bx = std::max(1, std::min(64, tile_h / 4));
    dim3 grid(bx, n_tiles);
    if (tile_dtype == SQ_U16)
        hipLaunchKernelGGL(minmax_kernel<uint16_t>, grid, dim3(256), 0, s, tile_ptrs_dev, tile_base_dev, tile_stride, tile_h,
                           tile_w, tile_pitch, out_minmax_dev);
    else
        hipLaunchKernelGGL(minmax_kernel<uint8_t>, grid, dim3(256), 0, s, tile_ptrs_dev, tile_base_dev, tile_stride, tile_h,
                           tile_w, tile_pitch, out_minmax_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_tile_minmax: launch failed: %s", hipGetErrorString(e));
    return SQ_OK;
}

extern "C" int sq_normalize_tiles(const void *const *tile_ptrs_dev, const void *tile_base_dev, int64_t tile_stride,
                                  int32_t n_tiles, int32_t tile_h, int32_t tile_w, int32_t tile_pitch, int32_t tile_dtype,
                                  const uint32_t *minmax_dev, void *out_dev, void *stream_) {
    if ((!tile_ptrs_dev && !tile_base_dev) || !minmax_dev || !out_dev || n_tiles < 0 || tile_h <= 0 || tile_w <= 0 ||
        tile_pitch < tile_w)
        return fail(SQ_ERR_INVALID, "sq_normalize_tiles: bad arguments (n_tiles=%d %dx%d pitch %d)", n_tiles, tile_h, tile_w,
                    tile_pitch);
    if (tile_dtype != SQ_U8 && tile_dtype != SQ_U16) return fail(SQ_ERR_UNSUPPORTED, "sq_normalize_tiles: dtype %d", tile_dtype);
    if (n_tiles == 0) return SQ_OK;
    if (n_tiles > 65535) return fail(SQ_ERR_UNSUPPORTED, "sq_normalize_tiles: more than 65535 tiles per call");
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>(1024, ((int64_t)tile_h * tile_w + 1023) / 1024));
    dim3 grid(bx, n_tiles);
    if (tile_dtype == SQ_U16)
        hipLaunchKernelGGL(normalize_kernel<uint16_t>, grid, dim3(256), 0, s, tile_ptrs_dev, tile_base_dev, tile_stride, tile_h,
                           tile_w, tile_pitch, minmax_dev, static_cast<uint16_t *>(out_dev));
    else
        hipLaunchKernelGGL(normalize_kernel<uint8_t>, grid, dim3(256), 0, s, tile_ptrs_dev, tile_base_dev, tile_stride, tile_h,
                           tile_w, tile_pitch, minmax_dev, static_cast<uint8_t *>(out_dev));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_normalize_tiles: launch failed: %s", hipGetErrorString(e));
    return SQ_OK;
}

extern "C" int sq_register_line_supported(int32_t n) { return line_supported(n) ? 1 : 0; }

extern "C" int sq_selftest_normalise_divide(uint64_t *mismatches_dev, void *stream_) {
    if (!mismatches_dev) return fail(SQ_ERR_INVALID, "sq_selftest_normalise_divide: NULL argument");
    hipStream_t s = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(selftest_normalise_kernel, dim3(65536), dim3(256), 0, s, reinterpret_cast<unsigned long long *>(mismatches_dev));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_selftest_normalise_divide: %s", hipGetErrorString(e));
    return SQ_OK;
}

extern "C" int64_t sq_register_workspace_bytes(int32_t n_pairs, int32_t n0, int32_t n1, int32_t upsample_factor) {
    if (n_pairs < 0 || n0 < 2 || n1 < 2 || upsample_factor < 1 || upsample_factor > 100)
        return fail(SQ_ERR_INVALID, "sq_register_workspace_bytes: bad sizes (pairs=%d crop=%dx%d u=%d)", n_pairs, n0, n1,
                    upsample_factor);
    int rc;
    if ((rc = check_line(n0, "axis-0")) != SQ_OK) return rc;
    if ((rc = check_line(n1, "axis-1")) != SQ_OK) return rc;
    return make_layout(n_pairs, n0, n1, upsample_factor).total;
}

extern "C" int sq_register_pairs(const sq_register_args *a, void *stream_) {
    if (!a || (!a->tile_ptrs_dev && !a->tile_base_dev) || !a->minmax_dev || !a->pairs_dev || !a->results_dev ||
        !a->workspace_dev)
        return fail(SQ_ERR_INVALID, "sq_register_pairs: NULL argument");
    if (a->n_pairs < 0 || a->n_pairs > 65535) return fail(SQ_ERR_INVALID, "sq_register_pairs: n_pairs %d out of range", a->n_pairs);
    if (a->upsample_factor < 1 || a->upsample_factor > 100)
        return fail(SQ_ERR_INVALID, "sq_register_pairs: upsample_factor %d out of range", a->upsample_factor);
    if (a->normalization != SQ_NORM_NONE && a->normalization != SQ_NORM_PHASE)
        return fail(SQ_ERR_INVALID, "normalization must be either phase or None (got %d)", a->normalization);
    if (a->tile_dtype != SQ_U8 && a->tile_dtype != SQ_U16) return fail(SQ_ERR_UNSUPPORTED, "sq_register_pairs: dtype %d", a->tile_dtype);
    if (a->n0 > a->tile_h || a->n1 > a->tile_w || a->tile_pitch < a->tile_w)
        return fail(SQ_ERR_INVALID, "sq_register_pairs: crop %dx%d larger than the %dx%d tile", a->n0, a->n1, a->tile_h, a->tile_w);
    int rc;
    if ((rc = check_line(a->n0, "axis-0")) != SQ_OK) return rc;
    if ((rc = check_line(a->n1, "axis-1")) != SQ_OK) return rc;
    const Layout L = make_layout(a->n_pairs, a->n0, a->n1, a->upsample_factor);
    if (a->workspace_bytes < L.total)
        return fail(SQ_ERR_WORKSPACE, "sq_register_pairs: workspace %lld < %lld bytes", (long long)a->workspace_bytes, (long long)L.total);
    if (reinterpret_cast<uintptr_t>(a->workspace_dev) % 16) return fail(SQ_ERR_INVALID, "sq_register_pairs: workspace not 16-byte aligned");
    if (a->n_pairs == 0) return SQ_OK;

    RegParams P{};
    P.tile_ptrs = a->tile_ptrs_dev;
    P.tile_base = a->tile_base_dev;
    P.tile_stride = a->tile_stride;
    P.tile_pitch = a->tile_pitch;
    P.minmax = a->minmax_dev;
    P.pairs = a->pairs_dev;
    P.results = a->results_dev;
    P.ws = static_cast<char *>(a->workspace_dev);
    P.L = L;
    P.n_pairs = a->n_pairs;
    P.normalization = a->normalization;
    P.n_tiles = a->n_tiles;
    P.tile_h = a->tile_h;
    P.tile_w = a->tile_w;
    // columns per block (directly transformed axis 0): two [tc][n0] complex arrays in LDS; a column too long for two to fit
    // (n0 > 4608), or one that needs a Bluestein line, goes one per block (columns_single_kernel).  SMALL blocks: a block
    // runs its phases one after the other (strided load, transforms, product, store, inverse, store), and only other
    // blocks of the CU can fill the gaps -- 16 KiB of columns per block (2 x 256 or, for longer columns, a single one)
    // with 256 threads measured 9.5 / 10.0 ms per 992-pair batch of 256- / 1024-point columns against 11.2 / 11.2 ms
    // with 128 KiB / 512 threads (profiles/r03_exp_registration_shapes.log).  A power of two, so that 8 / tc blocks
    // share the spectra's 128-byte lines (P.share).
    int tc = 0;
    if (!L.m0 && !L.long0 && 2 * (int64_t)L.n0 * 16 <= 144 * 1024) {
        tc = 1;
        while (tc < 8 && 2 * (int64_t)(2 * tc) * L.n0 * 16 <= 16 * 1024) tc *= 2;
        // ... but enough lines that a stage has a butterfly for every thread: n0 / r of them per line, r the largest radix
        int rmax = 4;
        for (int i = 0; i < L.ax0.nf; ++i) rmax = std::max<int>(rmax, L.ax0.radix[i]);
        while (tc < 8 && 2 * (int64_t)tc * (L.n0 / rmax) < 256 && 2 * (int64_t)(2 * tc) * L.n0 * 16 <= 144 * 1024) tc *= 2;
    }
    int col_threads = 2 * (int64_t)std::max(tc, 1) * L.n0 * 16 >= 64 * 1024 ? SQ_COL_THREADS : 256;   // a long column brings its own waves
#ifdef SQ_EXPERIMENTS      // (the product library reads no environment variable)
    if (const char *e = getenv("SQ_REG_TC")) tc = tc ? std::max(1, std::min(atoi(e), (int)((144 * 1024) / (2 * (int64_t)L.n0 * 16)))) : 0;
    if (const char *e = getenv("SQ_REG_COL_THREADS")) col_threads = std::min(SQ_COL_THREADS, std::max(64, atoi(e)));
#endif
    P.tc = tc;
    hipStream_t s = static_cast<hipStream_t>(stream_);

    hipLaunchKernelGGL(init_tables_kernel, dim3(64), dim3(256), 0, s, P);
    if ((L.m0 && !L.long0) || (L.m1 && !L.long1)) {   // spectra of the Bluestein chirps, one block per axis
        const size_t lds_chirp = (size_t)std::max(L.long0 ? 0 : L.m0, L.long1 ? 0 : L.m1) * 16;
        if ((rc = allow_lds(init_chirp_kernel<false>, lds_chirp)) != SQ_OK) return rc;
        hipLaunchKernelGGL(init_chirp_kernel<false>, dim3(2), dim3(256), lds_chirp, s, P, -1);
    }
    for (int axis = 0; axis < 2; ++axis)      // ... of a long axis: in place in the workspace
        if ((axis ? L.m1 : L.m0) && (axis ? L.long1 : L.long0))
            hipLaunchKernelGGL(init_chirp_kernel<true>, dim3(1), dim3(1024), 0, s, P, axis);
    // Lines per block of the row kernels, measured on 240-pair batches: the forward kernel (global loads
    // + a float64 normalisation per pixel) likes many small blocks -- 16 KB of lines; the inverse kernel
    // (LDS FFT + a per-wave argmax) likes up to 8 lines within 64 KB.  Never more than keeps ~2 blocks
    // per CU busy when the batch is small (the bench's single centre pairs).
    const int64_t line_bytes = (int64_t)(L.m1 ? L.m1 : L.n1) * 16;   // a Bluestein line is m1 points long
    const int64_t dft_scratch = 0;
    auto lines_per_block = [&](int cap, int n_lines) {
        if (L.long1) return 1;
        int rl = (int)std::max<int64_t>(1, std::min<int64_t>(cap, (64 * 1024 - dft_scratch) / line_bytes));
        while (rl > 1 && (int64_t)a->n_pairs * ((n_lines + rl - 1) / rl) < 512) rl >>= 1;
        return rl;
    };
    int rlf = lines_per_block((int)std::max<int64_t>(1, std::min<int64_t>(8, 16384 / line_bytes)), L.n0);
    int rli = lines_per_block((int)std::max<int64_t>(1, std::min<int64_t>(8, 32768 / line_bytes)), (L.n0 + 1) / 2);
#ifdef SQ_EXPERIMENTS
    if (const char *e = getenv("SQ_REG_RLF")) rlf = std::max(1, std::min<int>(atoi(e), (int)(160 * 1024 / line_bytes)));
    if (const char *e = getenv("SQ_REG_RLI")) rli = std::max(1, std::min<int>(atoi(e), (int)(160 * 1024 / line_bytes)));
#endif
    P.rl_fwd = rlf;
    P.rl_inv = rli;
    // one LONG line per block (a Bluestein line of thousands of points: up to 152 KB of LDS, so one or two blocks per CU):
    // the block brings the waves that hide its latencies itself -- 1024 threads from 32 KB of line on, 512 from 16 KB
    // (measured on 312 x 3122 crops, lines of 6400 points: rows forward 2.87 -> see profiles/r03_kernel_probe_registration.log)
    auto line_threads = [&](int rl) {
        if (rl > 1) return 256;
        // (a 1024-point line on its own, 16 KB exactly: 256 threads = one radix-4 butterfly each per pass, 8.05 against
        // 8.45 ms per 992-pair batch with 512, profiles/r03_exp_registration_threads.log)
        return line_bytes >= 32 * 1024 ? 1024 : (line_bytes > 16 * 1024 ? 512 : pick_threads(L.n1));
    };
    int ntf = line_threads(rlf), nti = line_threads(rli);
#ifdef SQ_EXPERIMENTS
    if (const char *e = getenv("SQ_REG_FWD_THREADS")) ntf = std::max(64, std::min(1024, atoi(e)));
    if (const char *e = getenv("SQ_REG_INV_THREADS")) nti = std::max(64, std::min(1024, atoi(e)));
#endif
    const size_t lds_fwd = L.long1 ? 0 : (size_t)(rlf * line_bytes + dft_scratch), lds_inv = L.long1 ? 0 : (size_t)(rli * line_bytes + dft_scratch);
    // the general (mixed-radix) instantiations only where an axis' plan has mixed-radix stages: see lines_fft_plan
    const bool gen0 = L.ax0.nf > 0, gen1 = L.ax1.nf > 0;
#define SQ_LAUNCH(KERNEL, GRID, THREADS, LDS)                                        \
    do {                                                                             \
        if ((rc = allow_lds(KERNEL, LDS)) != SQ_OK) return rc;                       \
        hipLaunchKernelGGL(KERNEL, GRID, dim3(THREADS), LDS, s, P);                  \
    } while (0)
    const dim3 grid_fwd((L.n0 + rlf - 1) / rlf, a->n_pairs);
    // a long axis: LONG_SLOTS workgroups (fewer when there are fewer lines), each with its scratch line, walk over the lines
    auto grid_long = [&](int64_t lines) { return dim3((unsigned)std::min<int64_t>(L.n_slots, lines)); };
    if (L.long1) {
        const dim3 g = grid_long((int64_t)a->n_pairs * L.n0);
        if (a->tile_dtype == SQ_U16) {
            if (gen1) SQ_LAUNCH((rows_forward_kernel<uint16_t, true, true>), g, LONG_THREADS, 0);
            else SQ_LAUNCH((rows_forward_kernel<uint16_t, false, true>), g, LONG_THREADS, 0);
        } else {
            if (gen1) SQ_LAUNCH((rows_forward_kernel<uint8_t, true, true>), g, LONG_THREADS, 0);
            else SQ_LAUNCH((rows_forward_kernel<uint8_t, false, true>), g, LONG_THREADS, 0);
        }
    } else if (a->tile_dtype == SQ_U16) {
        if (gen1) SQ_LAUNCH((rows_forward_kernel<uint16_t, true>), grid_fwd, ntf, lds_fwd);
        else SQ_LAUNCH((rows_forward_kernel<uint16_t, false>), grid_fwd, ntf, lds_fwd);
    } else {
        if (gen1) SQ_LAUNCH((rows_forward_kernel<uint8_t, true>), grid_fwd, ntf, lds_fwd);
        else SQ_LAUNCH((rows_forward_kernel<uint8_t, false>), grid_fwd, ntf, lds_fwd);
    }
    if (L.long0) {
        const dim3 g = grid_long((int64_t)a->n_pairs * L.n1h);
        if (gen0) SQ_LAUNCH((columns_single_kernel<true, true>), g, LONG_THREADS, 0);
        else SQ_LAUNCH((columns_single_kernel<false, true>), g, LONG_THREADS, 0);
    } else if (tc < 1) {
        const size_t lds_col = (size_t)(L.m0 ? L.m0 : L.n0) * 16;
        P.share = 8;
        const dim3 grid_col((L.n1h + 7) / 8 * 8, a->n_pairs);
        const int ntc = lds_col > 80 * 1024 ? 1024 : SQ_COL_THREADS;     // one block per CU: twice the waves
        if (gen0) SQ_LAUNCH(columns_single_kernel<true>, grid_col, ntc, lds_col);
        else SQ_LAUNCH(columns_single_kernel<false>, grid_col, ntc, lds_col);
    } else {
        const size_t lds_col = (size_t)2 * tc * L.n0 * 16;
        const int share = (tc == 1 || tc == 2 || tc == 4) ? 8 / tc : 1;
        P.share = share;
        const dim3 grid_col(((L.n1h + tc - 1) / tc + share - 1) / share * share, a->n_pairs);
        if (gen0) SQ_LAUNCH(columns_kernel<true>, grid_col, col_threads, lds_col);
        else SQ_LAUNCH(columns_kernel<false>, grid_col, col_threads, lds_col);
    }
    const dim3 grid_inv(((L.n0 + 1) / 2 + rli - 1) / rli, a->n_pairs);
    if (L.long1) {
        const dim3 g = grid_long((int64_t)a->n_pairs * ((L.n0 + 1) / 2));
        if (gen1) SQ_LAUNCH((rows_inverse_kernel<true, true>), g, LONG_THREADS, 0);
        else SQ_LAUNCH((rows_inverse_kernel<false, true>), g, LONG_THREADS, 0);
    } else if (gen1) SQ_LAUNCH(rows_inverse_kernel<true>, grid_inv, nti, lds_inv);
    else SQ_LAUNCH(rows_inverse_kernel<false>, grid_inv, nti, lds_inv);
#undef SQ_LAUNCH
    hipLaunchKernelGGL(peak_kernel, dim3(a->n_pairs), dim3(256), 0, s, P);
    if (a->upsample_factor > 1) {
        const int row_pairs = (L.n0 + 1) / 2;    // slots of a row and its mirror (upsample_rows_kernel)
        if ((int64_t)a->n_pairs * ((row_pairs + 63) / 64) >= 256)
            hipLaunchKernelGGL((upsample_rows_kernel<4, 16>), dim3((row_pairs + 63) / 64, a->n_pairs), dim3(256), 0, s, P);
        else
            hipLaunchKernelGGL((upsample_rows_kernel<1, 32>), dim3((row_pairs + 15) / 16, a->n_pairs), dim3(256), 0, s, P);
        hipLaunchKernelGGL(upsample_cols_kernel, dim3(L.region, a->n_pairs), dim3(256), 0, s, P);
        hipLaunchKernelGGL(upsample_peak_kernel, dim3(a->n_pairs), dim3(256), 0, s, P);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_register_pairs: launch failed: %s", hipGetErrorString(e));
    return SQ_OK;
}
