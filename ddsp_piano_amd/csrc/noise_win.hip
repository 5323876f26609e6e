// Time-varying FilteredNoise FIR for gfx950, "one lane = one frame" form (round 3).
//
// Same operator as noise.hip (ddsp.core.frequency_filter as reached from DynamicSizeFilteredNoise.get_signal,
// ddsp_piano/modules/filtered_noise_synth.py:27-42):
//   out[n] = sum_j noise[j] * ir_{j / U}[n + delay - j]
// What changed is who owns what.  noise.hip's kernel gives a lane 16 consecutive outputs and a quarter of the
// input blocks that reach them: the 19 taps of a step are re-read from LDS at every step (six ds_read_b128 per
// 64 FMAs) and the LDS pipe, not the VALU, is the busy unit.  Here the 64 lanes of a
// wavefront are 32 consecutive FRAMES x 2 output phases: lane (fr, ph) owns the OPL outputs
// U (F0 + fr) + OPL ph .. + OPL - 1 and walks ALL the input blocks that reach them, one block (4 samples) per step,
// from the latest block to the earliest.  Every lane of a half-wave is at the same position relative to its own
// frame grid, so
//   * the tap indices of a step are the same for all lanes (each lane reads them from its OWN frame's impulse
//     response): from one step to the next the window of OPL + 3 taps slides by four -- the window lives in a
//     register ring and a step loads ONE new 16-byte tap block and ONE noise block: two ds_read_b128 per 4 OPL
//     FMAs instead of six per 64;
//   * a lane changes frame only where its block index crosses a multiple of U / 4 (twice per walk): the walk is
//     cut into per-frame segments, each starting with a fresh ring (all control flow is wave-uniform);
//   * there is nothing to add up across lanes: no shuffles, no segment bookkeeping, one accumulation chain per
//     output in a fixed order (the two-call form runs the same core: bit-identical results).
// LDS layout: images of D = 32 frames at stride gs floats (gs / 4 odd: the 16 lanes of a ds_read_b128 service group
// are 16 consecutive frames and hit 16 different bank quads), zero gaps between the images wide enough for every
// tap index a step can form (no bounds logic in the loop); the noise of the D frames with one pad block per frame
// (stride U / 4 + 1 blocks, odd).  A window computes W = D - 2 frames (the first and the last designed frame only
// feed their neighbours).
#include "ddspp_common.h"
#include <stdio.h>

#include "noise_win.h"

// DDSPP_WIN_DEBUG's timing ablations (bits 0 .. 7: no walk, no design, priorities, a shorter walk -- deliberately WRONG audio)
// exist only in -DDDSPP_ABLATIONS builds (tools/build_variant.py); in the shipped library the kernels see a constant 0
// there and the compiler removes the branches.  Bits 8 .. (the stride of the TRACE instrumentation) are not an ablation.
#ifdef DDSPP_ABLATIONS
#define DDSPP_WIN_ABLATION_BITS(d) (d)
#else
#define DDSPP_WIN_ABLATION_BITS(d) ((d) & ~0xff)
#endif

namespace ddspp {

constexpr int WIN_D = 32;            // designed frames per window = two MFMA row tiles

__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// magnitude tile [D][K + 4]: rows padded by four floats.  The matrix-core A-operand read is a ds_read_b64 per lane,
// serviced in two groups of 32 lanes = 16 rows x 2 adjacent float pairs, bank = dword address mod 64: with a row stride
// of K + 4 = 4 or 36 (mod 64) for K = 32 / 64 / 96 / 128 the sixteen rows start 4 banks apart in some order and the 64
// dwords of a group fall on 64 different banks -- conflict free (rounds 3-4: unpadded rows with their 8-float groups
// XOR-swizzled by (s / 2) % 4, rows s and s + 8 on the same banks: SQ_LDS_BANK_CONFLICT 16.4 M cycles per launch, 18 %
// of the LDS cycles).  512 bytes more per workgroup: 53 664 at the headline shape, still three workgroups per CU.
constexpr int WIN_MPAD = 4;

// One lane's walk, frame by frame (g = frame offset relative to the lane's own frame, from the latest frame that
// reaches the lane's outputs to the earliest).  The two half-waves are two adjacent output phases A and B = A + 1:
// B's block range is OPL / 4 steps higher.  Both halves walk the UNION of the two ranges: the steps a half does not
// need only meet taps outside [0, Lw) -- zeros of the gaps between the frame images -- so every loop bound and every
// frame change is wave-uniform, nothing is masked, and a lane only adds its frame's offset to the two LDS pointers
// (54 issued steps for 51 at the headline shape).
//   gl: the lane's tap pointer for (g = 0, q = 0): image of its own frame + padl + OPL ph + delay - 3
//   xl: the lane's noise pointer for (g = 0, r = 0)
//   q_hi, q_lo: first (highest) block index of phase B, last (lowest) of phase A, relative to the lane's frame.
// HEAD / TAIL (round 4): position of the step in the walk when it is one of the first OPL / 4 + 1 or last OPL / 4 steps
// of a trimmed walk (WinGeom::trim), else -1.  Both half-waves walk the union of their block ranges, so at the top of
// the walk only phase B meets taps at all, at the bottom only phase A, and in those steps most (tap, sample) pairs lie
// outside [0, Lw) for BOTH halves: with m = e - d,
//   step i from the top      : tap_B = OPL - 1 - ... >= 0   <=>  m >= OPL - 1 - 4 i        (tap_A = tap_B - OPL < 0 there)
//   step j from the bottom   : tap_A <= Lw - 1              <=>  m <= 4 j                  (tap_B = tap_A + OPL > Lw - 1)
// (for the shapes win_geometry marks: (delay + OPL - 1) % 4 == 0 and Lw + 2 + OPL == 4 nsteps -- every shipped one).  The
// pairs outside only ever multiply zeros of the gaps between the frame images: leaving them out changes no result
// and saves 168 of the 2 592 multiply-adds of a walk at the headline shape.
template <int OPL, int AQ, int RS, int HEAD = -1, int TAIL = -1>
__device__ __forceinline__ void fir_win_step(const float4 (&ring)[RS], const float4& xq, int u, float (&acc)[OPL]) {
    float tp[4 * AQ];
#pragma unroll
    for (int k = 0; k < AQ; ++k) {
        const float4 t = ring[(u + k) % RS];
        tp[4 * k] = t.x; tp[4 * k + 1] = t.y; tp[4 * k + 2] = t.z; tp[4 * k + 3] = t.w;
    }
    const float xs[4] = {xq.x, xq.y, xq.z, xq.w};
    constexpr int M_LO = HEAD >= 0 ? OPL - 1 - 4 * HEAD : -4, M_HI = TAIL >= 0 ? 4 * TAIL : OPL;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < OPL; ++e)
            if (e - d >= M_LO && e - d <= M_HI) acc[e] = __builtin_fmaf(xs[d], tp[e - d + 3], acc[e]);
}

template <int OPL, int AQ, int RS, bool IS_HEAD, int POS = 0>
__device__ __forceinline__ void fir_win_step_at(const float4 (&ring)[RS], const float4& xq, int u, int pos, float (&acc)[OPL]) {
    if constexpr (POS <= OPL / 4) {
        if (pos == POS) {
            if (IS_HEAD) fir_win_step<OPL, AQ, RS, POS, -1>(ring, xq, u, acc);
            else fir_win_step<OPL, AQ, RS, -1, POS>(ring, xq, u, acc);
        } else {
            fir_win_step_at<OPL, AQ, RS, IS_HEAD, POS + 1>(ring, xq, u, pos, acc);
        }
    }
}

// One whole turn of the ring: RS steps, the tap block and the noise block of step u + 2 issued BEFORE step u's FMAs.
// KIND 0: plain; 1: the walk's first turn (steps 0 .. OPL / 4 trimmed from the top); 2: its last (the last OPL / 4 steps).
template <int OPL, int AQ, int RS, int KIND>
__device__ __forceinline__ void fir_win_turn(float4 (&ring)[RS], float4 (&xr)[RS], const float* gp, const float* xb,
                                             float (&acc)[OPL]) {
    constexpr int HN = OPL / 4 + 1, TN = OPL / 4;
#pragma unroll
    for (int u = 0; u < RS; ++u) {
        ring[(u + AQ + 1) % RS] = lds4(gp + 4 * (u + AQ + 1));
        xr[(u + 2) % RS] = lds4(xb + 4 * (RS - u - 2));
        __builtin_amdgcn_sched_barrier(0);
        if (KIND == 1 && u < HN) fir_win_step_at<OPL, AQ, RS, true>(ring, xr[u % RS], u, u, acc);
        else if (KIND == 2 && u >= RS - TN) fir_win_step_at<OPL, AQ, RS, false>(ring, xr[u % RS], u, RS - 1 - u, acc);
        else fir_win_step<OPL, AQ, RS>(ring, xr[u % RS], u, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int OPL, int BPF>
__device__ __forceinline__ void fir_win_core(const float* __restrict__ gl, const float* __restrict__ xl, int q_hi,
                                             int q_lo, int gs, float (&acc)[OPL], bool trim = false) {
    constexpr int AQ = (OPL + 6) / 4;        // 16-byte blocks holding the OPL + 3 taps of a step
    constexpr int RS = AQ + 2;               // register ring: the step's window + the blocks of the next two steps
    const int g_hi = q_hi >= 0 ? q_hi / BPF : -((-q_hi + BPF - 1) / BPF);
    const int g_lo = q_lo >= 0 ? q_lo / BPF : -((-q_lo + BPF - 1) / BPF);
    for (int g = g_hi; g >= g_lo; --g) {
        const int fq0 = g * BPF;
        const int qs = min(q_hi, fq0 + BPF - 1), len = qs - max(q_lo, fq0) + 1;
        const float* gp = gl + g * gs - 4 * qs;
        const float* xb = xl + 4 * ((BPF + 1) * g + qs - fq0) - 4 * RS;    // noise blocks are walked downwards
        float4 ring[RS], xr[RS];
#pragma unroll
        for (int k = 0; k <= AQ; ++k) ring[k] = lds4(gp + 4 * k);
        xr[0] = lds4(xb + 4 * RS);
        xr[1] = lds4(xb + 4 * (RS - 1));
        // the walk's first steps open its first segment, its last steps close its last one; they are trimmed where they
        // fall into whole turns (first segment of at least a turn; last segment a whole number of turns) -- true for every
        // wavefront at the 24 kHz and 48 kHz shapes, for some at the others: an untrimmed step is merely not shortened
        const bool head = trim && g == g_hi && g != g_lo && len >= RS;
        const bool tail = trim && g == g_lo && g != g_hi && len % RS == 0;
        int i0 = 0;
        const int len_t = tail ? len - RS : len;
        if (head) {
            fir_win_turn<OPL, AQ, RS, 1>(ring, xr, gp, xb, acc);
            asm volatile("; walk: first turn");
            gp += 4 * RS; xb -= 4 * RS; i0 = RS;
        }
        for (; i0 + RS <= len_t; i0 += RS) {                             // whole turns of the ring: no exits inside
            fir_win_turn<OPL, AQ, RS, 0>(ring, xr, gp, xb, acc);
            gp += 4 * RS;
            xb -= 4 * RS;
        }
        if (tail) {                                                      // (then the segment has no rest)
            fir_win_turn<OPL, AQ, RS, 2>(ring, xr, gp, xb, acc);
            asm volatile("; walk: last turn");
            i0 += RS;
        }
        // the rest of the segment: nested tests -- a segment of whole turns (every one at the 24 kHz shape) leaves at the
        // first (they were RS - 1 separate tests, each a scalar branch: fifteen per walk for nothing)
        [&] {
#pragma unroll
            for (int u = 0; u < RS - 1; ++u) {
                if (i0 + u >= len) return;
                ring[(u + AQ + 1) % RS] = lds4(gp + 4 * (u + AQ + 1));
                xr[(u + 2) % RS] = lds4(xb + 4 * (RS - u - 2));
                __builtin_amdgcn_sched_barrier(0);
                fir_win_step<OPL, AQ, RS>(ring, xr[u % RS], u, acc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }();
    }
}

// ------------------------------------------------------------------------------------------------
// The walk on the matrix pipe (round 4).  A convolution is a sum of outer products: with A = four consecutive taps and
// B = four consecutive input samples, v_mfma_f32_4x4x1_16b_f32 adds SIXTEEN independent 4 x 4 outer products -- one per
// block of four lanes -- in one instruction: 256 multiply-adds, every one of them a (tap, sample) pair the filter needs.
// The matrix pipe sustains the same multiply-add rate as the vector pipe on paper (tools/ubench/mfma_ceiling: 72-74
// TMAC/s), but at 1050 W instead of the 1350-1400 W at which the vector walk is throttled, with a quarter of the
// instructions, and without the vector walk's waste (two halves of a wavefront walking the union of their ranges, the
// ragged ends of 12-output lanes).
//   lane block b = frame fr = 16 fh + b of the window (fh = which half of its 32 frames), sub = lane & 3;
//   accumulator c (4 registers): D_c[i][jj] = sum over input blocks q of  tap[4 t + rho + i] * x[4 q + jj],
//   t = c + dq - q, which is a term of output n_rel = 4 c + (i + jj) of the frame (rho = delay % 4, dq = delay / 4);
//   a wavefront owns QB consecutive c: the QB - 1 output quads c0 + 1 .. c0 + QB - 1, whose values are the (i + jj < 4)
//   part of D_c plus the (i + jj >= 4) part of D_{c-1} (quad_perm adds, win_mfma_quad below).
// Per input block (one step): ONE new tap block joins a ring of QB + 1 registers, one x block is loaded, and up to QB
// instructions are issued -- those whose tap block lies in [t_min, t_max] (a bit mask: the first and the last QB - 1
// steps of a walk are triangles, and a pair outside the range would read the neighbouring image).  As in fir_win_core
// the walk is cut where the input block changes frame (the taps are those of the INPUT sample's frame): a fresh ring per
// segment, every loop bound wave-uniform.
typedef float win_f4 __attribute__((ext_vector_type(4)));
constexpr int win_floordiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// The walk is SKEWED: at step sg every accumulator k meets the SAME tap block t = t_min + sg and its own input block
// q = Q0 + k - sg (Q0 = c0 + dq - t_min), so that all QB instructions of all NT = t_max - t_min + 1 steps are pairs the
// filter needs -- no triangles at the ends, no masks, no loop, no branch: NT steps of straight-line code.  The QB input
// blocks of a step are consecutive; one new block per step joins a ring of QB + 1 registers.  The taps are those of the
// INPUT block's frame: the QB blocks of a step lie in at most two frames (QB <= BPF), so a step fetches the tap block
// from the image of one or two frames.  Everything about the schedule -- which accumulator takes which image at which
// step, which input blocks lie outside the staged frames (they only feed pairs no output of the task uses, but an
// instruction computes all sixteen products: they enter as zeros) -- follows from (hop, delay, taps, task) and is
// resolved at COMPILE time (DELAY, LW, RL, RH, OH are template arguments, checked against the geometry at launch): a
// first version that decided them at run time spent its time in scalar branches (0.79 ms), a second one in 637 selects
// with 1 700 spilled scalar registers.
constexpr int WIN_MW_AHEAD = 2;       // steps whose operands are already requested (one step is ~110 cycles of the matrix pipe)
// What rides in the walk's instruction stream (fir_win_mfma's last argument): nothing, or the FIR design of the NEXT unit
// (WinDesignRide below: two 16 x 16 x 4 matrix instructions and one LDS read of magnitudes per walk step, 24 steps).
struct WinNoRide {
    template <int SG>
    __device__ __forceinline__ void step() {}
};
template <int QB, int BPF, int T_MIN, int NT, int Q0, int Q_MIN, int Q_MAX, int SG>
struct WinMfmaSteps {
    static constexpr int RSZ = QB + WIN_MW_AHEAD;
    // the operands of step S: its new input block into the ring slot no accumulator holds then, its one or two tap blocks
    template <int S>
    static __device__ __forceinline__ void fetch(const float* __restrict__ gi, const float* __restrict__ xl, int gs,
                                                 float (&xr)[RSZ], float& tlo, float& thi) {
        if constexpr (S < NT) {
            constexpr int qn = Q0 - S;
            constexpr int slot = ((-S) % RSZ + RSZ) % RSZ;
            if constexpr (S > 0) {                                        // (step 0's blocks are loaded with the ring)
                if constexpr (qn >= Q_MIN && qn <= Q_MAX) xr[slot] = xl[4 * (qn + win_floordiv(qn, BPF))];
                else xr[slot] = 0.0f;
            }
            constexpr int n_lo = win_floordiv(qn, BPF), n_hi = win_floordiv(qn + QB - 1, BPF);
            tlo = gi[n_lo * gs + 4 * (T_MIN + S)];
            if constexpr (n_hi != n_lo) thi = gi[n_hi * gs + 4 * (T_MIN + S)];
            else thi = tlo;
        }
    }
    // tl[j] / th[j]: the tap blocks of step SG + j
    template <class Ride>
    static __device__ __forceinline__ void run(const float* __restrict__ gi, const float* __restrict__ xl, int gs,
                                               win_f4 (&acc)[QB], float (&xr)[RSZ], float (&tl)[WIN_MW_AHEAD], float (&th)[WIN_MW_AHEAD],
                                               Ride& ride) {
        if constexpr (SG < NT) {
            constexpr int g_lo = win_floordiv(Q0 - SG, BPF);
            const float ta = tl[0], tb = th[0];
#pragma unroll
            for (int j = 0; j + 1 < WIN_MW_AHEAD; ++j) {
                tl[j] = tl[j + 1];
                th[j] = th[j + 1];
            }
            fetch<SG + WIN_MW_AHEAD>(gi, xl, gs, xr, tl[WIN_MW_AHEAD - 1], th[WIN_MW_AHEAD - 1]);
            __builtin_amdgcn_sched_barrier(0);                // (one basic block of NT steps: keep the loads where they are)
#pragma unroll
            for (int k = 0; k < QB; ++k) {
                const float a = win_floordiv(Q0 + k - SG, BPF) == g_lo ? ta : tb;
                acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, xr[((k - SG) % RSZ + RSZ) % RSZ], acc[k], 0, 0, 0);
            }
            ride.template step<SG>();
            __builtin_amdgcn_sched_barrier(0);
            WinMfmaSteps<QB, BPF, T_MIN, NT, Q0, Q_MIN, Q_MAX, SG + 1>::run(gi, xl, gs, acc, xr, tl, th, ride);
        }
    }
};

template <int QB, int BPF, int DELAY, int LW, int RL, int RH, int OH, class Ride>
__device__ __forceinline__ void fir_win_mfma(const float* __restrict__ gi,    // lane: image of its own frame + padl + rho + sub
                                             const float* __restrict__ xl,    // lane: noise of its own frame + sub
                                             int gs, win_f4 (&acc)[QB], Ride& ride) {
    constexpr int RHO = DELAY & 3, DQ = DELAY >> 2, T_MIN = RHO == 0 ? 0 : -1, T_MAX = (LW - 1 - RHO) >> 2;
    constexpr int NT = T_MAX - T_MIN + 1;
    constexpr int Q_MIN = -RL * BPF, Q_MAX = (RH + 1) * BPF - 1;
    constexpr int C0 = (QB - 1) * OH - 1, Q0 = C0 + DQ - T_MIN;
    static_assert(QB <= BPF, "the input blocks of a step must not span three frames");
    typedef WinMfmaSteps<QB, BPF, T_MIN, NT, Q0, Q_MIN, Q_MAX, 0> Steps;
    // block q of frame g lies at xl + 4 ((BPF + 1) g + q - g BPF) = xl + 4 (q + g)
    float xr[Steps::RSZ];
#pragma unroll
    for (int k = 0; k < QB; ++k) {
        const int q = Q0 + k;
        xr[k] = (q >= Q_MIN && q <= Q_MAX) ? xl[4 * (q + win_floordiv(q, BPF))] : 0.0f;
    }
    // (template recursion, not `#pragma unroll`: the second instance of the NT x QB body was left a loop -- and a ring
    // indexed by a loop variable lives in scratch memory: 2.4 ms)
    float tl[WIN_MW_AHEAD], th[WIN_MW_AHEAD];
    Steps::template fetch<0>(gi, xl, gs, xr, tl[0], th[0]);
    Steps::template fetch<1>(gi, xl, gs, xr, tl[1], th[1]);
    static_assert(WIN_MW_AHEAD == 2, "the prologue fetches steps 0 and 1");
    Steps::run(gi, xl, gs, acc, xr, tl, th, ride);
}
// output quad c of a lane block from the accumulators of c and c - 1: lane sub gets n_rel = 4 c + sub
template <int CTRL>
__device__ __forceinline__ float win_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float win_mfma_quad(const win_f4& lower, const win_f4& own, int sub) {
    const float w0 = own[0];
    const float w1 = (sub + 1 >= 4) ? lower[1] : own[1];
    const float w2 = (sub + 2 >= 4) ? lower[2] : own[2];
    const float w3 = (sub + 3 >= 4) ? lower[3] : own[3];
    return ((w0 + win_dpp<0x93>(w1)) + win_dpp<0x4E>(w2)) + win_dpp<0x39>(w3);     // quad rotations by 1, 2, 3 lanes
}

// the walk of one task and its QB - 1 output quads: o[c] = output 4 (c0 + c) + sub of the lane block's frame, c = 1 .. QB - 1
template <int QB, int BPF, int DELAY, int LW, int RL, int RH, int OH, class Ride>
__device__ __forceinline__ void fir_win_mfma_quads(const float* __restrict__ gi, const float* __restrict__ xl, int gs, int sub,
                                                   float (&o)[QB], Ride& ride) {
    win_f4 acc[QB];
#pragma unroll
    for (int c = 0; c < QB; ++c) acc[c] = win_f4{0.f, 0.f, 0.f, 0.f};
    fir_win_mfma<QB, BPF, DELAY, LW, RL, RH, OH>(gi, xl, gs, acc, ride);
#pragma unroll
    for (int c = 1; c < QB; ++c) o[c] = win_mfma_quad(acc[c - 1], acc[c], sub);
}

// per-lane pointers of the walk for output phase ph of window-relative frame fr
template <int OPL, int BPF>
__device__ __forceinline__ void win_lane(const WinGeom& g, const float* G, const float* Xs, int fr, int ph,
                                         const float*& gl, const float*& xl) {
    gl = G + (fr + g.RL) * g.gs + g.padl + OPL * ph + g.delay - 3;
    xl = Xs + 4 * (BPF + 1) * (fr + g.RL);
}

// noise of the D frames of a window -> registers (clamped addresses), registers -> LDS (zero outside the signal)
template <int BPF, int XQ>
__device__ __forceinline__ void win_fetch_x(const float* __restrict__ xrow, int nblk, int jb0, float4 (&xv)[XQ]) {
    const float4* xg = reinterpret_cast<const float4*>(xrow);
#pragma unroll
    for (int u = 0; u < XQ; ++u) xv[u] = xg[min(max(jb0 + (int)threadIdx.x + 256 * u, 0), nblk - 1)];
}
// ... or drawn in place (WinGeom::draw_on): the same clamped blocks, from the counters of their place in a [R, N] tensor
template <int BPF, int XQ>
__device__ __forceinline__ void win_draw_x(unsigned long long seed, unsigned long long row_base, int nblk, int jb0,
                                           float4 (&xv)[XQ]) {
#pragma unroll
    for (int u = 0; u < XQ; ++u) {
        const int b = threadIdx.x + 256 * u;
        if (XQ * 256 == BPF * WIN_D || b < BPF * WIN_D)
            xv[u] = philox_uniform4(seed, row_base + (unsigned long long)min(max(jb0 + b, 0), nblk - 1));
    }
}
template <int BPF, int XQ>
__device__ __forceinline__ void win_store_x(float* __restrict__ Xs, int nblk, int jb0, const float4 (&xv)[XQ]) {
#pragma unroll
    for (int u = 0; u < XQ; ++u) {
        const int b = threadIdx.x + 256 * u, jb = jb0 + b;
        if (b < BPF * WIN_D)
            *reinterpret_cast<float4*>(Xs + 4 * (b + b / BPF)) =
                (jb >= 0 && jb < nblk) ? xv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// FilteredNoise in one kernel: FIR design on the matrix cores (E = M_even CE, O = M_odd CO, tap weights straight
// from the accumulators into the frame images in LDS) + the walk above.  Persistent workgroups, three barriers
// per window of W frames (noise.hip: three per 1024 outputs).
// ------------------------------------------------------------------------------------------------
// TRACE (tools/ubench/noise_win_trace.hip only): every wavefront of the first workgroups writes the clock at its phase
// boundaries to `trace`.
constexpr int WIN_TRACE_WGS = 64, WIN_TRACE_UNITS = 8, WIN_TRACE_MARKS = 8;
template <int KH, int JT, int OPL, int BPF, bool TRACE, int QB = 0, int MW_DELAY = 0, int MW_LW = 0, int MW_RL = 0, int MW_RH = 0>
__device__ __forceinline__ void
noise_win_fused_body(const float* __restrict__ x,          // [R, N] noise
                       const float* __restrict__ mags,       // [R, T, 2 KH]
                       const float* __restrict__ CE, const float* __restrict__ CO,     // [KH, NJ]
                       const int* __restrict__ tap_idx, const float* __restrict__ tap_we,
                       const float* __restrict__ tap_wo,     // [NJ, 4]
                       float* __restrict__ out,              // [R / vq, N]
                       float* __restrict__ out_last,         // [R / n_voices, N] or null
                       int R, int N, int T, int NJ, WinGeom g, float bias, ScaleFn scale, int vq, int n_voices,
                       int vmajor, int tpw, int dbg_arg, long long* __restrict__ trace) {
    const int dbg = DDSPP_WIN_ABLATION_BITS(dbg_arg);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int K = 2 * KH, KS = KH / 4, D = WIN_D, U = 4 * BPF, NP = U / OPL, NPASS = (NP + 7) / 8;
    constexpr int XQ = (BPF * D + 255) / 256, PER_ROW = K / 4, MQ = (D * PER_ROW + 255) / 256;
    constexpr int TW = 4 / JT;                                   // wavefront groups that share the tiles of a window
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    // LDS: [magnitudes D x (K + 4): padded rows][padded noise of D frames][D frame images].  Nothing valid ever reads below
    // the first image's tap 0 or above the last image's last tap, so the first image starts `gshift` floats early
    // (its lower gap overlaps the noise region) and there is no tail: 53 664 bytes at the headline shape -- three
    // workgroups per CU (the allocation granule makes 53 888 bytes two).
    constexpr int MS = K + WIN_MPAD;                          // floats per row of the magnitude tile
    float* M = lds_dyn;                                       // [D][MS]
    float* Xs = M + D * MS;                                   // padded noise of D frames
    float* Gtop = Xs + (BPF + 1) * 4 * D;                     // first float of the image region
    float* G = Gtop - g.gshift;                               // image s, tap k: G[s gs + padl + k]
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wibs = wave_uniform(wib);
    const int col = lane & 15, kq = lane >> 4;

    // design role: wavefront (tg, jt) owns the 16-column block jt of the tiles tg, tg + TW, ...; its table fragments
    // stay in registers, the tap weights of its column (12 values) are re-read per window (first-level cache hits)
    const bool designer = wib < JT * TW;
    const int jt = wib % JT, tg = wib / JT;
    const int jcol = 16 * jt + col;
    const int jc = min(jcol, NJ - 1);
    float bE[KS], bO[KS];
#pragma unroll
    for (int st = 0; st < KS; ++st) {
        bE[st] = CE[(4 * st + kq) * NJ + jc];
        bO[st] = CO[(4 * st + kq) * NJ + jc];
    }
    for (int i = threadIdx.x; i < D * g.gs - g.gshift; i += 256) Gtop[i] = 0.0f;      // the gaps stay zero for ever
    // The table fragments must have LANDED before the loop: with loads pending at its entry the compiler makes every
    // trip wait for "all loads" in the middle of the design (that is, for the noise prefetch: ~2 us per unit).
#pragma unroll
    for (int st = 0; st < KS; ++st) asm volatile("" ::"v"(bE[st]), "v"(bO[st]));
    // FIR role: lane = (frame fr, phase 2 wib + half) of every pass
    const int half = lane >> 5, fr = min(lane & 31, g.W - 1);
    const bool fr_ok = (lane & 31) < g.W;

    const int nblk = N / 4;
    const int ntasks = (R / vq) * g.wpr;
    const int n_seg = R / n_voices, pq = n_voices / vq;
    const int nunits = ntasks * vq;                               // (task, voice) units, walked voice-fastest

    auto unit_geometry = [&](int unit, int& row, int& F0) {
        const int task = unit / vq, iv = unit - task * vq;
        const int orow = task / g.wpr;
        if (vq == 1) {
            row = orow;
        } else {
            const int b = orow / pq, v = (orow - b * pq) * vq + iv;
            row = vmajor ? v * n_seg + b : b * n_voices + v;
        }
        F0 = (task - orow * g.wpr) * g.W;
    };
    float4 xv[XQ], mv[MQ];
    const int upw = tpw * vq;
    const int u_begin = min((int)blockIdx.x * upw, nunits), u_end = min(u_begin + upw, nunits);
    // The tap weights of a column are 12 values (4 indices, 4 even and 4 odd weights) and the four lanes (col, kq = 0..3)
    // of a column need the same twelve: each keeps three of them for the life of the workgroup and the design fetches
    // the other nine with ds_bpermute -- 3 registers instead of 12, and no loads in the loop (columns past NJ repeat
    // column NJ - 1: same values, same places).
    int tq[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int v = 3 * kq + i;                                  // 0..3 index, 4..7 even weight, 8..11 odd weight
        const int* src = v < 4 ? tap_idx : (v < 8 ? reinterpret_cast<const int*>(tap_we) : reinterpret_cast<const int*>(tap_wo));
        tq[i] = src[4 * jc + (v & 3)];
    }
    asm volatile("" ::"v"(tq[0]), "v"(tq[1]), "v"(tq[2]));          // landed before the loop (see above)
    int4 ti;
    float4 we, wo;
    auto load_taps = [&]() {
        int v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = __shfl(tq[i % 3], col + 16 * (i / 3));
        ti = make_int4(v[0], v[1], v[2], v[3]);
        we = make_float4(__int_as_float(v[4]), __int_as_float(v[5]), __int_as_float(v[6]), __int_as_float(v[7]));
        wo = make_float4(__int_as_float(v[8]), __int_as_float(v[9]), __int_as_float(v[10]), __int_as_float(v[11]));
    };
    auto fetch_x = [&](int unit) {
        int row, F0;
        unit_geometry(min(unit, nunits - 1), row, F0);
        if (g.draw_on) {                 // (wave-uniform)
            if (unit < u_end)            // nothing is drawn for a unit no walk will read
                win_draw_x<BPF, XQ>(g.draw_seed, g.draw_offset + (unsigned long long)row * nblk, nblk, BPF * (F0 - g.RL), xv);
        } else {
            win_fetch_x<BPF, XQ>(x + (size_t)row * N, nblk, BPF * (F0 - g.RL), xv);
        }
    };
    auto store_x = [&](int unit) {
        int row, F0;
        unit_geometry(min(unit, nunits - 1), row, F0);
        win_store_x<BPF, XQ>(Xs, nblk, BPF * (F0 - g.RL), xv);
    };
    auto fetch_m = [&](int unit) {
        int row, F0;
        unit_geometry(min(unit, nunits - 1), row, F0);
#pragma unroll
        for (int u = 0; u < MQ; ++u) {
            const int i = min((int)threadIdx.x + 256 * u, D * PER_ROW - 1);
            const int s = i / PER_ROW, c4 = i - s * PER_ROW;
            const int f = min(max(F0 - g.RL + s, 0), T - 1);
            mv[u] = reinterpret_cast<const float4*>(mags + ((size_t)row * T + f) * K)[c4];
        }
    };
    auto store_m = [&]() {                                        // scale_fn on raw magnitudes on the way
        with_scale_kind(scale.kind, [&](auto kind) {              // (the kind decided once, not per magnitude)
#pragma unroll
            for (int u = 0; u < MQ; ++u) {
                const int i = threadIdx.x + 256 * u;
                if (i < D * PER_ROW) {
                    const float4 m = scale4_of<decltype(kind)::value>(scale, mv[u], bias);
                    const int s = i / PER_ROW, c4 = i - s * PER_ROW;
                    *reinterpret_cast<float4*>(M + s * MS + 4 * c4) = m;
                }
            }
        });
    };
    // A workgroup walks a contiguous run of tpw tasks (tpw vq units); two barriers per unit:
    //   [M(u) in LDS]  design u (M -> G) -> BARRIER -> fetch M(u+1), walk u (G, Xs), M(u+1) to LDS -> BARRIER ->
    //   noise(u+1) registers -> LDS, fetch noise(u+2)
    // Both prefetches have a whole walk to land (measured: with the magnitudes fetched behind the design only, the
    // kernel waited ~2 us per unit for them).
    if (u_begin < u_end) {
        fetch_x(u_begin);
        fetch_m(u_begin);
        store_x(u_begin);
        store_m();
        fetch_x(u_begin + 1);
    }
    __syncthreads();
    float vsum[NPASS][OPL];
    // matrix-pipe walk (QB > 0): a wavefront's task = (half of the window's 32 frames, group of QB - 1 output quads);
    // the accumulators run on through the voices of a sum
    constexpr int MW_NPH = QB > 0 ? U / (4 * (QB > 0 ? QB - 1 : 1)) : 1, MW_PASS = QB > 0 ? (2 * MW_NPH + 3) / 4 : 1;
    constexpr int QBA = QB > 0 ? QB : 1;
    float mvs[MW_PASS][QBA];      // sums over the voices of a row's outputs (the vector walk's vsum)
    const int tstride = TRACE ? max(dbg >> 8, 1) : 1;            // TRACE: every tstride-th workgroup is recorded
    auto mark = [&](int unit, int k) {
        if (TRACE && (int)blockIdx.x % tstride == 0 && (int)blockIdx.x / tstride < WIN_TRACE_WGS &&
            unit - u_begin < WIN_TRACE_UNITS && lane == 0)
            trace[(((size_t)(blockIdx.x / tstride) * WIN_TRACE_UNITS + (unit - u_begin)) * 4 + wib) * WIN_TRACE_MARKS + k] =
                wall_clock64();
    };
    for (int unit = u_begin; unit < u_end; ++unit) {
        const int task = unit / vq, iv = unit - task * vq;
        int row, F0;
        unit_geometry(unit, row, F0);
        mark(unit, 0);
        // ---- 1. E / O blocks on the matrix cores; tap weights -> frame images (lane holds E, O of the frames
        // 4 kq .. 4 kq + 3 of the tile at column 16 jt + col)
        if (dbg & 4) __builtin_amdgcn_s_setprio(3);
        if (designer && !(dbg & 2)) {
            load_taps();
            for (int t = tg; t < D / 16; t += TW) {
                f32x4 accE = f32x4{0.f, 0.f, 0.f, 0.f}, accO = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* arow = M + (16 * t + col) * MS + 2 * kq;
#pragma unroll
                for (int st = 0; st < KS; ++st) {
                    const float2 am = *reinterpret_cast<const float2*>(arow + 8 * st);
                    accE = __builtin_amdgcn_mfma_f32_16x16x4f32(am.x, bE[st], accE, 0, 0, 0);
                    accO = __builtin_amdgcn_mfma_f32_16x16x4f32(am.y, bO[st], accO, 0, 0, 0);
                }
                // An unused slot has index -1 and weights 0: it writes a zero to the gap float below tap 0 -- no branches.
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float E = accE[rr], O = accO[rr];
                    float* dst = G + (16 * t + 4 * kq + rr) * g.gs + g.padl;
                    dst[ti.x] = __builtin_fmaf(wo.x, O, we.x * E);
                    dst[ti.y] = __builtin_fmaf(wo.y, O, we.y * E);
                    dst[ti.z] = __builtin_fmaf(wo.z, O, we.z * E);
                    dst[ti.w] = __builtin_fmaf(wo.w, O, we.w * E);
                }
            }
        }
        mark(unit, 1);
        __syncthreads();
        mark(unit, 2);
        fetch_m(unit + 1);               // in flight during the walk (the design alone is too short to hide the latency)
        if (dbg & 4) __builtin_amdgcn_s_setprio(0);
        // ---- 2. the walk: NPASS passes of 8 phases
        const int orow = task / g.wpr;
        const bool lastv = out_last != nullptr && iv == vq - 1 && (orow % pq) == pq - 1;
        if constexpr (QB > 0) {
            const int sub = lane & 3, blk = lane >> 2;
#pragma unroll
            for (int p = 0; p < MW_PASS; ++p) {
                const int tk = 4 * p + wibs;                             // (wave-uniform)
                if (tk < 2 * MW_NPH) {
                    const int fh = tk & 1, oh = tk >> 1;
                    const int frw = 16 * fh + blk;
                    float o[QBA];                                        // this voice's outputs 4 c + sub of the task's quads
#pragma unroll
                    for (int c = 1; c < QB; ++c) o[c] = 0.0f;
                    if (!(dbg & 1)) {
                        // (the 4 QB accumulator registers live only inside the call: held across the voices of a sum
                        // they did not fit beside the design's fragments -- 79 spilled registers)
                        const int frc = min(frw, g.W - 1);
                        const float* gi = G + (frc + g.RL) * g.gs + g.padl + (MW_DELAY & 3) + sub;
                        const float* xl = Xs + 4 * (BPF + 1) * (frc + g.RL) + sub;
                        static_assert(QB == 0 || MW_NPH <= 2, "one instance of the walk per group of output quads");
                        WinNoRide none;
                        if (oh == 0) fir_win_mfma_quads<QBA, BPF, MW_DELAY, MW_LW, MW_RL, MW_RH, 0>(gi, xl, g.gs, sub, o, none);
                        else fir_win_mfma_quads<QBA, BPF, MW_DELAY, MW_LW, MW_RL, MW_RH, 1>(gi, xl, g.gs, sub, o, none);
                    }
                    if (iv == 0) {
#pragma unroll
                        for (int c = 1; c < QB; ++c) mvs[p][c] = o[c];
                    } else if (!lastv) {
#pragma unroll
                        for (int c = 1; c < QB; ++c) mvs[p][c] += o[c];
                    }
                    if (iv == vq - 1 && frw < g.W && F0 + frw < T) {
                        const size_t n = (size_t)U * (F0 + frw) + 4 * (QB - 1) * oh + sub;
                        float* dst = out + (size_t)orow * N + n;
#pragma unroll
                        for (int c = 1; c < QB; ++c) dst[4 * (c - 1)] = mvs[p][c];
                        if (lastv) {
                            float* dl = out_last + (size_t)(orow / pq) * N + n;
#pragma unroll
                            for (int c = 1; c < QB; ++c) dl[4 * (c - 1)] = o[c];
                        }
                    }
                }
            }
        } else {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int ph = 8 * p + 2 * wib + half;
            float acc[OPL];
#pragma unroll
            for (int e = 0; e < OPL; ++e) acc[e] = 0.f;
            if ((NP % 8 == 0 || 8 * p + 2 * wibs < NP) && !(dbg & 1)) {   // wave-uniform (NP is even)
                const float *gl, *xl;
                win_lane<OPL, BPF>(g, G, Xs, fr, min(ph, NP - 1), gl, xl);
                const int qA = g.q_hi0 + (OPL / 4) * (8 * p + 2 * wibs);     // phase A's first block
                const int cut = (dbg & 8) ? OPL / 4 : 0;         // timing only (wrong audio): what would 11 % fewer steps buy?
                fir_win_core<OPL, BPF>(gl, xl, qA + OPL / 4 - cut, qA - g.nsteps + 1 + cut, g.gs, acc,
                                       g.trim != 0 && !(dbg & 16));
            }
            // out_last: the segment's last voice leaves on its own and stays out of the sum -- the outputs dictionary
            // of the reference's DAG holds that voice's noise next to the mix
            if (iv == 0) {
#pragma unroll
                for (int e = 0; e < OPL; ++e) vsum[p][e] = acc[e];
            } else if (!lastv) {
#pragma unroll
                for (int e = 0; e < OPL; ++e) vsum[p][e] += acc[e];
            }
            if (iv == vq - 1 && fr_ok && ph < NP && F0 + fr < T) {
                const size_t n = (size_t)U * (F0 + fr) + OPL * ph;
                float4* o = reinterpret_cast<float4*>(out + (size_t)orow * N + n);
#pragma unroll
                for (int e4 = 0; e4 < OPL / 4; ++e4)
                    o[e4] = make_float4(vsum[p][4 * e4], vsum[p][4 * e4 + 1], vsum[p][4 * e4 + 2], vsum[p][4 * e4 + 3]);
                if (lastv) {
                    float4* ol = reinterpret_cast<float4*>(out_last + (size_t)(orow / pq) * N + n);
#pragma unroll
                    for (int e4 = 0; e4 < OPL / 4; ++e4)
                        ol[e4] = make_float4(acc[4 * e4], acc[4 * e4 + 1], acc[4 * e4 + 2], acc[4 * e4 + 3]);
                }
            }
        }
        }
        mark(unit, 3);
        if (dbg & 4) __builtin_amdgcn_s_setprio(3);
        store_m();                       // M has been free since the design
        mark(unit, 4);
        __syncthreads();                 // G and Xs are free; M holds the next unit's magnitudes
        mark(unit, 5);
        store_x(unit + 1);
        fetch_x(unit + 2);
        mark(unit, 6);
    }
}

// ------------------------------------------------------------------------------------------------
// The design RIDING in the matrix-pipe walk (round 4, DDSPP_WIN_MFMA=2).  With the walk on the matrix pipe the kernel
// is no longer bound by the walk but by how little of a workgroup's design / staging overlaps another workgroup's walk
// (DESIGN.md 5a).  Here the 48 matrix instructions that design unit u + 1 are dealt out behind the first 24 steps of
// the walk of unit u (WinDesignRide), their accumulators stay in registers until unit u's images are dead, and the
// magnitudes run one unit further ahead.  Per trip:
//   images u (design registers -> G), noise u (registers -> Xs), magnitudes u + 1 (registers -> M), fetch noise u + 1 and
//   magnitudes u + 2   -> BARRIER ->   walk u with the design of u + 1 riding, outputs   -> BARRIER
// Two wavefronts per SIMD (the walk's 52 accumulators, the design's 16, the table fragments, two prefetches).
// (Also built: the noise / magnitude staging riding in the walk as well, double-buffered in 78 KB of LDS.  Same results;
// 0.575 ms against this kernel's 0.54: a wavefront's instruction stream is serial, what rides in it delays its own
// matrix instructions, and two wavefronts per SIMD do not cover for each other.)
// ------------------------------------------------------------------------------------------------
template <int KS>
struct WinDesignRide {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    float bE[KS], bO[KS];          // table fragments of this wavefront's column block
    f32x4 hE[2], hO[2];            // E / O of the two row tiles of the next unit
    const float* mlane;            // M + col MS + 2 kq (row tile 0)
    int K;                         // floats per row of the magnitude tile (MS)
    float2 am;                     // magnitudes of the next ride step
    __device__ __forceinline__ void reset() {
        hE[0] = hE[1] = hO[0] = hO[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        am = *reinterpret_cast<const float2*>(mlane);
    }
    template <int SG>
    __device__ __forceinline__ void step() {
        if constexpr (SG < 2 * KS) {
            constexpr int t = SG / KS, st = SG % KS;
            const float2 a = am;
            if constexpr (SG + 1 < 2 * KS) {
                constexpr int tn = (SG + 1) / KS, sn = (SG + 1) % KS;
                am = *reinterpret_cast<const float2*>(mlane + 16 * tn * K + 8 * sn);
            }
            hE[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bE[st], hE[t], 0, 0, 0);
            hO[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bO[st], hO[t], 0, 0, 0);
        }
    }
    __device__ __forceinline__ void all() {          // the design standing alone (a workgroup's first unit)
        reset();
        run_all<0>();
    }
    template <int SG>
    __device__ __forceinline__ void run_all() {
        if constexpr (SG < 2 * KS) {
            step<SG>();
            run_all<SG + 1>();
        }
    }
};

template <int KH, int JT, int BPF, int QB, int DELAY, int LW, int RL, int RH, bool TRACE>
__device__ __forceinline__ void
noise_win_ride_body(const float* __restrict__ x, const float* __restrict__ mags, const float* __restrict__ CE,
                    const float* __restrict__ CO, const int* __restrict__ tap_idx, const float* __restrict__ tap_we,
                    const float* __restrict__ tap_wo, float* __restrict__ out, float* __restrict__ out_last, int R, int N, int T,
                    int NJ, WinGeom g, float bias, ScaleFn scale, int vq, int n_voices, int vmajor, int tpw, int dbg_arg,
                    long long* __restrict__ trace) {
    const int dbg = DDSPP_WIN_ABLATION_BITS(dbg_arg);
    constexpr int K = 2 * KH, KS = KH / 4, D = WIN_D, U = 4 * BPF;
    constexpr int XQ = (BPF * D + 255) / 256, PER_ROW = K / 4, MQ = (D * PER_ROW + 255) / 256;
    static_assert(JT == 3 && D == 32, "wavefronts 0 .. 2 design both row tiles of their column block, wavefront 3 none");
    static_assert(U == 8 * (QB - 1), "two groups of QB - 1 output quads per frame");
    static_assert(2 * KS <= (LW + 2) / 4, "the design's steps must fit the walk's");
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    constexpr int MS = K + WIN_MPAD;                          // floats per row of the magnitude tile
    float* M = lds_dyn;                                       // [D][MS]
    float* Xs = M + D * MS;                                   // padded noise of D frames
    float* Gtop = Xs + (BPF + 1) * 4 * D;
    float* G = Gtop - g.gshift;                               // image s, tap k: G[s gs + padl + k]
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wibs = wave_uniform(wib);
    const int col = lane & 15, kq = lane >> 4;
    const bool designer = wibs < JT;                          // wavefront jt designs column block jt of both row tiles
    const int jcol = 16 * min(wib, JT - 1) + col;
    const int jc = min(jcol, NJ - 1);
    WinDesignRide<KS> ride;
#pragma unroll
    for (int st = 0; st < KS; ++st) {
        ride.bE[st] = CE[(4 * st + kq) * NJ + jc];
        ride.bO[st] = CO[(4 * st + kq) * NJ + jc];
    }
    ride.mlane = M + col * MS + 2 * kq;
    ride.K = MS;
    for (int i = threadIdx.x; i < D * g.gs - g.gshift; i += 256) Gtop[i] = 0.0f;      // the gaps stay zero for ever
#pragma unroll
    for (int st = 0; st < KS; ++st) asm volatile("" ::"v"(ride.bE[st]), "v"(ride.bO[st]));
    const int nblk = N / 4;
    const int ntasks = (R / vq) * g.wpr;
    const int n_seg = R / n_voices, pq = n_voices / vq;
    const int nunits = ntasks * vq;
    auto unit_geometry = [&](int unit, int& row, int& F0) {
        const int task = unit / vq, iv = unit - task * vq;
        const int orow = task / g.wpr;
        if (vq == 1) {
            row = orow;
        } else {
            const int b = orow / pq, v = (orow - b * pq) * vq + iv;
            row = vmajor ? v * n_seg + b : b * n_voices + v;
        }
        F0 = (task - orow * g.wpr) * g.W;
    };
    float4 xv[XQ], mv[MQ];
    int tq[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int v = 3 * kq + i;                                  // 0..3 index, 4..7 even weight, 8..11 odd weight
        const int* src = v < 4 ? tap_idx : (v < 8 ? reinterpret_cast<const int*>(tap_we) : reinterpret_cast<const int*>(tap_wo));
        tq[i] = src[4 * jc + (v & 3)];
    }
    asm volatile("" ::"v"(tq[0]), "v"(tq[1]), "v"(tq[2]));
    auto fetch_x = [&](int unit) {
        int row, F0;
        unit_geometry(min(unit, nunits - 1), row, F0);
        win_fetch_x<BPF, XQ>(x + (size_t)row * N, nblk, BPF * (F0 - g.RL), xv);
    };
    auto store_x = [&](int unit) {
        int row, F0;
        unit_geometry(min(unit, nunits - 1), row, F0);
        win_store_x<BPF, XQ>(Xs, nblk, BPF * (F0 - g.RL), xv);
    };
    auto fetch_m = [&](int unit) {
        int row, F0;
        unit_geometry(min(unit, nunits - 1), row, F0);
#pragma unroll
        for (int u = 0; u < MQ; ++u) {
            const int i = min((int)threadIdx.x + 256 * u, D * PER_ROW - 1);
            const int sr = i / PER_ROW, c4 = i - sr * PER_ROW;
            const int f = min(max(F0 - g.RL + sr, 0), T - 1);
            mv[u] = reinterpret_cast<const float4*>(mags + ((size_t)row * T + f) * K)[c4];
        }
    };
    auto store_m = [&]() {                                        // scale_fn on raw magnitudes on the way
        with_scale_kind(scale.kind, [&](auto kind) {
#pragma unroll
            for (int u = 0; u < MQ; ++u) {
                const int i = threadIdx.x + 256 * u;
                if (i < D * PER_ROW) {
                    const float4 m = scale4_of<decltype(kind)::value>(scale, mv[u], bias);
                    const int sr = i / PER_ROW, c4 = i - sr * PER_ROW;
                    *reinterpret_cast<float4*>(M + sr * MS + 4 * c4) = m;
                }
            }
        });
    };
    // tap weights -> frame images (a lane holds E, O of the frames 4 kq .. 4 kq + 3 of a tile at column 16 jt + col).
    // An unused slot has index -1 and weights 0: it writes a zero to the gap float below tap 0 -- no branches.
    auto write_images = [&]() {
        if (!designer) return;
        int v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = __shfl(tq[i % 3], col + 16 * (i / 3));
        const int4 ti = make_int4(v[0], v[1], v[2], v[3]);
        const float4 we = make_float4(__int_as_float(v[4]), __int_as_float(v[5]), __int_as_float(v[6]), __int_as_float(v[7]));
        const float4 wo = make_float4(__int_as_float(v[8]), __int_as_float(v[9]), __int_as_float(v[10]), __int_as_float(v[11]));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float E = ride.hE[t][rr], O = ride.hO[t][rr];
                float* dst = G + (16 * t + 4 * kq + rr) * g.gs + g.padl;
                dst[ti.x] = __builtin_fmaf(wo.x, O, we.x * E);
                dst[ti.y] = __builtin_fmaf(wo.y, O, we.y * E);
                dst[ti.z] = __builtin_fmaf(wo.z, O, we.z * E);
                dst[ti.w] = __builtin_fmaf(wo.w, O, we.w * E);
            }
        }
    };
    const int upw = tpw * vq;
    const int u_begin = min((int)blockIdx.x * upw, nunits), u_end = min(u_begin + upw, nunits);
    if (u_begin >= u_end) return;
    fetch_x(u_begin);
    fetch_m(u_begin);
    store_m();                           // magnitudes of the first unit
    fetch_m(u_begin + 1);
    __syncthreads();
    if (designer) ride.all();            // the first unit's design stands alone
    __syncthreads();                     // M is free
    float mvs[QB];                       // sums over the voices of a row
    const int tstride = TRACE ? max(dbg >> 8, 1) : 1;
    auto mark = [&](int unit, int k) {   // 0 trip start | 1 images | 2 noise in LDS | 3 magnitudes in LDS | 4 fetches issued | 5 past the barrier | 6 walk done
        if (TRACE && (int)blockIdx.x % tstride == 0 && (int)blockIdx.x / tstride < WIN_TRACE_WGS &&
            unit - u_begin < WIN_TRACE_UNITS && lane == 0)
            trace[(((size_t)(blockIdx.x / tstride) * WIN_TRACE_UNITS + (unit - u_begin)) * 4 + wib) * WIN_TRACE_MARKS + k] =
                wall_clock64();
    };
    const int sub = lane & 3, blk = lane >> 2;
    const int fh = wibs & 1, oh = wibs >> 1;                     // the wavefront's task: half of the frames, half of the quads
    for (int unit = u_begin; unit < u_end; ++unit) {
        const int task = unit / vq, iv = unit - task * vq;
        int row, F0;
        unit_geometry(unit, row, F0);
        mark(unit, 0);
        write_images();
        mark(unit, 1);
        store_x(unit);
        mark(unit, 2);
        store_m();                       // magnitudes of unit + 1 (the ride reads them)
        mark(unit, 3);
        fetch_x(unit + 1);               // in flight during the walk
        fetch_m(unit + 2);
        mark(unit, 4);
        __syncthreads();
        mark(unit, 5);
        const int orow = task / g.wpr;
        const bool lastv = out_last != nullptr && iv == vq - 1 && (orow % pq) == pq - 1;
        const int frw = 16 * fh + blk;
        float o[QB];
#pragma unroll
        for (int c = 1; c < QB; ++c) o[c] = 0.0f;
        {
            const int frc = min(frw, g.W - 1);
            const float* gi = G + (frc + g.RL) * g.gs + g.padl + (DELAY & 3) + sub;
            const float* xl = Xs + 4 * (BPF + 1) * (frc + g.RL) + sub;
            if (designer) {
                ride.reset();
                if (oh == 0) fir_win_mfma_quads<QB, BPF, DELAY, LW, RL, RH, 0>(gi, xl, g.gs, sub, o, ride);
                else fir_win_mfma_quads<QB, BPF, DELAY, LW, RL, RH, 1>(gi, xl, g.gs, sub, o, ride);
            } else {
                WinNoRide none;
                fir_win_mfma_quads<QB, BPF, DELAY, LW, RL, RH, 1>(gi, xl, g.gs, sub, o, none);     // (wavefront 3: oh = 1)
            }
        }
        if (iv == 0) {
#pragma unroll
            for (int c = 1; c < QB; ++c) mvs[c] = o[c];
        } else if (!lastv) {
#pragma unroll
            for (int c = 1; c < QB; ++c) mvs[c] += o[c];
        }
        if (iv == vq - 1 && frw < g.W && F0 + frw < T) {
            const size_t n = (size_t)U * (F0 + frw) + 4 * (QB - 1) * oh + sub;
            float* dst = out + (size_t)orow * N + n;
#pragma unroll
            for (int c = 1; c < QB; ++c) dst[4 * (c - 1)] = mvs[c];
            if (lastv) {
                float* dl = out_last + (size_t)(orow / pq) * N + n;
#pragma unroll
                for (int c = 1; c < QB; ++c) dl[4 * (c - 1)] = o[c];
            }
        }
        mark(unit, 6);
        __syncthreads();                 // G, Xs and M are free
    }
}

// Three workgroups per CU (168 registers) for the shapes whose LDS allows it; K = 128 (70 KB of LDS: two workgroups per CU)
// gets the registers of two wavefronts per SIMD instead.
#define DDSPP_WIN_KERNEL_ARGS                                                                                          \
    const float* __restrict__ x, const float* __restrict__ mags, const float* __restrict__ CE,                         \
    const float* __restrict__ CO, const int* __restrict__ tap_idx, const float* __restrict__ tap_we,                   \
    const float* __restrict__ tap_wo, float* __restrict__ out, float* __restrict__ out_last, int R, int N, int T,     \
    int NJ, WinGeom g, float bias, ScaleFn scale, int vq, int n_voices, int vmajor, int tpw, int dbg,                 \
    long long* __restrict__ trace
template <int KH, int JT, int OPL, int BPF, bool TRACE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
noise_win_fused_kernel(DDSPP_WIN_KERNEL_ARGS = nullptr) {
    noise_win_fused_body<KH, JT, OPL, BPF, TRACE>(x, mags, CE, CO, tap_idx, tap_we, tap_wo, out, out_last, R, N, T, NJ, g, bias,
                                                  scale, vq, n_voices, vmajor, tpw, dbg, trace);
}
#ifndef DDSPP_WIN_MW_WAVES
#define DDSPP_WIN_MW_WAVES 3
#endif
template <int KH, int JT, int OPL, int BPF, int QB, int DELAY, int LW, int RL, int RH, bool TRACE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DDSPP_WIN_MW_WAVES, DDSPP_WIN_MW_WAVES)))
noise_win_fused_mw_kernel(DDSPP_WIN_KERNEL_ARGS = nullptr) {
    noise_win_fused_body<KH, JT, OPL, BPF, TRACE, QB, DELAY, LW, RL, RH>(x, mags, CE, CO, tap_idx, tap_we, tap_wo, out, out_last, R, N, T, NJ, g,
                                                      bias, scale, vq, n_voices, vmajor, tpw, dbg, trace);
}
template <int KH, int JT, int BPF, int QB, int DELAY, int LW, int RL, int RH, bool TRACE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
noise_win_fused_ride_kernel(DDSPP_WIN_KERNEL_ARGS = nullptr) {
    noise_win_ride_body<KH, JT, BPF, QB, DELAY, LW, RL, RH, TRACE>(x, mags, CE, CO, tap_idx, tap_we, tap_wo, out, out_last, R, N, T,
                                                                    NJ, g, bias, scale, vq, n_voices, vmajor, tpw, dbg, trace);
}
template <int KH, int JT, int OPL, int BPF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
noise_win_fused_w2_kernel(DDSPP_WIN_KERNEL_ARGS = nullptr) {
    noise_win_fused_body<KH, JT, OPL, BPF, false>(x, mags, CE, CO, tap_idx, tap_we, tap_wo, out, out_last, R, N, T, NJ, g, bias,
                                                  scale, vq, n_voices, vmajor, tpw, dbg, trace);
}
#undef DDSPP_WIN_KERNEL_ARGS

// ------------------------------------------------------------------------------------------------
// The two-call form's second half (impulse responses [R, T, Lw] from HBM): same layout, same walk.
// ------------------------------------------------------------------------------------------------
template <int OPL, int BPF>
__global__ void __launch_bounds__(256) tv_fir_win_kernel(const float* __restrict__ x,   // [R, N]
                                                       const float* __restrict__ ir,  // [R, T, Lw]
                                                       float* __restrict__ out,       // [R, N]
                                                       int R, int N, int T, int Lw, WinGeom g) {
    constexpr int D = WIN_D, U = 4 * BPF, NP = U / OPL, NPASS = (NP + 7) / 8, XQ = (BPF * D + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    float* Xs = lds_dyn;
    float* Gtop = Xs + (BPF + 1) * 4 * D;
    float* G = Gtop - g.gshift;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wibs = wave_uniform(wib);
    const int half = lane >> 5, fr = min(lane & 31, g.W - 1);
    const bool fr_ok = (lane & 31) < g.W;
    const int nblk = N / 4;
    const int ntasks = R * g.wpr;
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const int row = task / g.wpr, F0 = (task - row * g.wpr) * g.W;
        float4 xv[XQ];
        win_fetch_x<BPF, XQ>(x + (size_t)row * N, nblk, BPF * (F0 - g.RL), xv);
        for (int i = g.gshift + threadIdx.x; i < D * g.gs; i += 256) {
            const int s = i / g.gs, tap = i - s * g.gs - g.padl;
            const int f = min(max(F0 - g.RL + s, 0), T - 1);
            G[i] = (tap >= 0 && tap < Lw) ? ir[((size_t)row * T + f) * Lw + tap] : 0.0f;
        }
        win_store_x<BPF, XQ>(Xs, nblk, BPF * (F0 - g.RL), xv);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int ph = 8 * p + 2 * wib + half;
            if (NP % 8 == 0 || 8 * p + 2 * wib < NP) {
                float acc[OPL];
#pragma unroll
                for (int e = 0; e < OPL; ++e) acc[e] = 0.f;
                const float *gl, *xl;
                win_lane<OPL, BPF>(g, G, Xs, fr, min(ph, NP - 1), gl, xl);
                const int qA = g.q_hi0 + (OPL / 4) * (8 * p + 2 * wibs);     // phase A's first block
                fir_win_core<OPL, BPF>(gl, xl, qA + OPL / 4, qA - g.nsteps + 1, g.gs, acc, g.trim != 0);
                if (fr_ok && ph < NP && F0 + fr < T) {
                    float4* o = reinterpret_cast<float4*>(out + (size_t)row * N + (size_t)U * (F0 + fr) + OPL * ph);
#pragma unroll
                    for (int e4 = 0; e4 < OPL / 4; ++e4)
                        o[e4] = make_float4(acc[4 * e4], acc[4 * e4 + 1], acc[4 * e4 + 2], acc[4 * e4 + 3]);
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int floordiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
static int ceildiv(int a, int b) { return -floordiv(-a, b); }

int win_opl_for(int U) {
    switch (U) {
        case 32: return 4;
        case 64: return 8;
        case 96: return 12;
        case 128: return 16;
        case 192: return 12;
        default: return 0;
    }
}

bool win_geometry(int N, int T, int Lw, int delay, WinGeom* g) {
    if (N <= 0 || T <= 0 || N % T != 0 || N % 4 != 0 || delay < 0 || Lw <= 0) return false;
    const int U = N / T, OPL = win_opl_for(U);
    if (OPL == 0) return false;
    const int bpf = U / 4, NP = U / OPL, AQ = (OPL + 6) / 4;
    const int q_hi0 = floordiv(delay + OPL - 1, 4), q_lo0 = ceildiv(delay - Lw - 2, 4);
    const int RL = -floordiv(q_lo0, bpf), RH = floordiv(q_hi0 + (OPL / 4) * (NP - 1), bpf);
    const int W = WIN_D - RL - RH;
    if (RL < 0 || RH < 0 || W < 16) return false;
    // zero taps a walk meets below an image (phase A at phase B's first step) and above it (phase B at phase A's
    // last step); one more block above is prefetched and never used
    int padl = OPL + 3 - delay + 4 * q_hi0;
    while ((padl + delay - 3) % 4 != 0) ++padl;
    const int above = delay + OPL - 3 - 4 * q_lo0 + 4 * AQ - Lw;     // (the blocks prefetched past it may hold anything)
    const int gap = (padl > above ? padl : above);
    int gs = Lw + gap;
    while (gs % 4 != 0 || (gs / 4) % 2 == 0) ++gs;
    g->U = U; g->delay = delay; g->padl = padl; g->gs = gs; g->nsteps = q_hi0 - q_lo0 + 1; g->q_hi0 = q_hi0;
    g->gshift = padl & ~3;
    g->RL = RL; g->RH = RH; g->W = W; g->wpr = (T + W - 1) / W; g->opl = OPL; g->bpf = bpf;
    // phase B's first tap is the first step's only one and phase A's last tap the last step's only one: the walk's ends
    // are the two triangles fir_win_step's HEAD / TAIL leave out
    g->trim = (delay + OPL - 1) % 4 == 0 && Lw + 2 + OPL == 4 * g->nsteps;
    g->Lw = Lw;
    g->draw_on = 0; g->draw_seed = 0; g->draw_offset = 0;        // (launch_win_fused sets them)
    return true;
}

size_t win_lds_bytes(const WinGeom& g, int K_or_0) {
    size_t fl = (size_t)WIN_D * g.gs - g.gshift + (size_t)(g.bpf + 1) * 4 * WIN_D;
    if (K_or_0 > 0) fl += (size_t)WIN_D * (K_or_0 + WIN_MPAD);
    return fl * sizeof(float);
}

bool win_fused_supported(int N, int T, int K, int Lw, int delay, WinGeom* g) {
    if (!win_geometry(N, T, Lw, delay, g) || Lw != 2 * (K - 1)) return false;
    const int U = g->U;
    // U = 192 at K = 96 (BASELINE config 5, 48 kHz: two passes of eight phases, 24 noise registers) and K = 128 (70 KB of
    // LDS) are two-workgroups-per-CU shapes and get the registers of two wavefronts per SIMD; every other hop keeps
    // round 2's kernel
    const bool inst = (K == 96 && (U == 96 || U == 192)) || (K == 64 && (U == 64 || U == 96)) ||
                      (K == 32 && (U == 128 || U == 32)) || (K == 128 && U == 128);
    return inst && win_lds_bytes(*g, K) <= 80 * 1024;          // (two workgroups per CU at least)
}

bool win_tvfir_supported(int N, int T, int Lw, int delay, WinGeom* g) {
    return win_geometry(N, T, Lw, delay, g) && win_lds_bytes(*g, 0) <= 64 * 1024;
}

int launch_win_fused(const float* audio, const float* magnitudes, const float* CE, const float* CO, const int* tap_idx,
                     const float* tap_we, const float* tap_wo, float* out, float* out_last, int R, int N, int T, int K,
                     int NJ, const WinGeom& g_in, float bias, const ScaleFn& sf, int vq, int n_voices, int voice_major,
                     hipStream_t stream, bool draw, unsigned long long draw_seed, unsigned long long draw_offset) {
    WinGeom g = g_in;
    g.draw_on = draw ? 1 : 0;
    g.draw_seed = draw_seed;
    g.draw_offset = draw_offset;
    const long long tasks = (long long)(R / vq) * g.wpr;
    DDSPP_REQUIRE((long long)R * g.wpr < (1ll << 31), "frequency_filter_eo: too many tasks");
    const size_t lds = win_lds_bytes(g, K) + (size_t)ddspp_option_literal("DDSPP_WIN_LDS_PAD", 0);     // pad: fewer workgroups per CU (A/B)
    // About eight (window, voice) units per workgroup: the set-up (zeroed images, table fragments) is paid once per
    // eight, and there are several times more workgroups than the chip holds at once, so the last round of a launch
    // is spread over all CUs by the dispatcher instead of leaving a fixed assignment's stragglers.
    int tpw = ddspp_option_literal("DDSPP_WIN_UNITS_PER_WG", 8) / vq;
    if (tpw < 1) tpw = 1;
    while (tpw > 1 && tasks / tpw < 768) tpw >>= 1;          // few tasks (a single segment): one unit per workgroup
#ifdef DDSPP_ABLATIONS
    const int dbg = ddspp_option_literal("DDSPP_WIN_DEBUG", 0);     // timing ablations: 1 = no walk, 2 = no design, 8 = shorter walk, 16 = untrimmed walk
#else
    const int dbg = ddspp_option_literal("DDSPP_WIN_DEBUG", 0) & ~0xff;    // the shipped library: only the trace stride (bits 8 ..) is read
#endif
    if (dbg & (1 | 2 | 8)) {                                // (these three leave work out: the audio is WRONG by design)
        static bool warned = false;
        if (!warned) {
            warned = true;
            fprintf(stderr, "libddspp: DDSPP_WIN_DEBUG=%d is a TIMING ABLATION -- FilteredNoise output is deliberately wrong; "
                            "unset it for anything but a timing experiment\n", dbg);
        }
    }
    const dim3 grid((unsigned)((tasks + tpw - 1) / tpw)), block(256);
#define DDSPP_WIN_LAUNCH(KH, JT, OPL, BPF)                                                                        \
    hipLaunchKernelGGL((noise_win_fused_kernel<KH, JT, OPL, BPF>), grid, block, lds, stream, audio, magnitudes, CE, \
                       CO, tap_idx, tap_we, tap_wo, out, out_last, R, N, T, NJ, g, bias, sf, vq, n_voices, voice_major, tpw, dbg)
    // more than 64 KB of dynamic LDS has to be allowed per function AND per device (the attribute is cheap to set: no
    // per-process flag, which would leave every device but the first without it)
#define DDSPP_WIN_LAUNCH_W2(KH, JT, OPL, BPF)                                                                          \
    do {                                                                                                               \
        DDSPP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&noise_win_fused_w2_kernel<KH, JT, OPL, BPF>), \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));                  \
        hipLaunchKernelGGL((noise_win_fused_w2_kernel<KH, JT, OPL, BPF>), grid, block, lds, stream, audio, magnitudes, \
                           CE, CO, tap_idx, tap_we, tap_wo, out, out_last, R, N, T, NJ, g, bias, sf, vq, n_voices,    \
                           voice_major, tpw, dbg);                                                                     \
    } while (0)
    const int U = g.U;
    // the matrix-pipe walk needs the taps below index 0 that a misaligned delay makes it read to be zeros of a gap: any
    // image but the window's first, i.e. at least one frame of look-back
    // the matrix-pipe walk: its schedule is compiled for one (delay, taps, look-back, look-ahead); a misaligned delay makes
    // it read taps below index 0, which are zeros of a gap in any image but the window's first (RL >= 1)
    const bool mw = ddspp_option_literal("DDSPP_WIN_MFMA", 0) && g.padl >= 8 && !draw;      // opt-in: same time as the vector walk
#define DDSPP_WIN_LAUNCH_MW(KH, JT, OPL, BPF, QB, DELAY, LW, RL, RH)                                                        \
    hipLaunchKernelGGL((noise_win_fused_mw_kernel<KH, JT, OPL, BPF, QB, DELAY, LW, RL, RH>), grid, block, lds, stream, audio, \
                       magnitudes, CE, CO, tap_idx, tap_we, tap_wo, out, out_last, R, N, T, NJ, g, bias, sf, vq, n_voices,   \
                       voice_major, tpw, dbg)
    if (K == 96 && U == 96 && mw && ddspp_option_literal("DDSPP_WIN_MFMA", 0) == 2 && g.delay == 93 && g.Lw == 190 && g.RL == 1 && g.RH == 1)
        hipLaunchKernelGGL((noise_win_fused_ride_kernel<48, 3, 24, 13, 93, 190, 1, 1>), grid, block, lds, stream, audio, magnitudes,
                           CE, CO, tap_idx, tap_we, tap_wo, out, out_last, R, N, T, NJ, g, bias, sf, vq, n_voices, voice_major, tpw,
                           dbg);
    else if (K == 96 && U == 96 && mw && g.delay == 93 && g.Lw == 190 && g.RL == 1 && g.RH == 1)
        DDSPP_WIN_LAUNCH_MW(48, 3, 12, 24, 13, 93, 190, 1, 1);
    else if (K == 96 && U == 96) DDSPP_WIN_LAUNCH(48, 3, 12, 24);
    else if (K == 64 && U == 64) DDSPP_WIN_LAUNCH(32, 2, 8, 16);
    else if (K == 64 && U == 96) DDSPP_WIN_LAUNCH(32, 2, 12, 24);
    else if (K == 32 && U == 128) DDSPP_WIN_LAUNCH(16, 1, 16, 32);
    else if (K == 32 && U == 32) DDSPP_WIN_LAUNCH(16, 1, 4, 8);                 // ENSTDkCl-8kHz.gin
    else if (K == 128 && U == 128) DDSPP_WIN_LAUNCH_W2(64, 4, 16, 32);           // ENSTDkCl-32kHz.gin: 70 KB of LDS
    else if (K == 96 && U == 192) DDSPP_WIN_LAUNCH_W2(48, 3, 12, 48);            // 48 kHz (BASELINE config 5): 65 KB
    else DDSPP_REQUIRE(false, "frequency_filter_eo: no windowed kernel for K=%d U=%d", K, U);
#undef DDSPP_WIN_LAUNCH
#undef DDSPP_WIN_LAUNCH_W2
#undef DDSPP_WIN_LAUNCH_MW
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

int launch_win_tvfir(const float* audio, const float* ir, float* out, int R, int N, int T, int Lw, const WinGeom& g,
                     hipStream_t stream) {
    const long long tasks = (long long)R * g.wpr;
    DDSPP_REQUIRE(tasks < (1ll << 31), "time_varying_fir: too many tasks");
    const size_t lds = win_lds_bytes(g, 0);
    const dim3 grid((unsigned)tasks), block(256);
#define DDSPP_WIN_LAUNCH(OPL, BPF) \
    hipLaunchKernelGGL((tv_fir_win_kernel<OPL, BPF>), grid, block, lds, stream, audio, ir, out, R, N, T, Lw, g)
    switch (g.U) {
        case 32: DDSPP_WIN_LAUNCH(4, 8); break;
        case 64: DDSPP_WIN_LAUNCH(8, 16); break;
        case 96: DDSPP_WIN_LAUNCH(12, 24); break;
        case 128: DDSPP_WIN_LAUNCH(16, 32); break;
        case 192: DDSPP_WIN_LAUNCH(12, 48); break;
        default: DDSPP_REQUIRE(false, "time_varying_fir: no windowed kernel for U=%d", g.U);
    }
#undef DDSPP_WIN_LAUNCH
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

}  // namespace ddspp
