// mdstep.hip -- gfx950 (MI355X / CDNA4) kernels + C-ABI launchers of the batched MetaDrive step().
//
// One workgroup (256 threads = 4 wave64) owns one environment for the whole step; phases are
// separated by workgroup barriers and never talk to other workgroups, so a step is ONE launch with
// no inter-workgroup traffic, no atomics and no grid-level synchronisation:
//
//   reset -> IDM(+trigger) -> integrate -> localize -> contacts -> traffic removal -> observe -> lidar
//
// Work shapes (wave64-native, none of this is a warp-32 tiling):
//   * lidar     : one wave per (agent, 64-beam sector).  Each LANE first loads ONE mover record
//                 (32 B, coalesced), culls it against range + the sector's cone, the survivors are
//                 collected with a 64-bit ballot and broadcast lane->wave with v_readlane; each lane
//                 then tests ITS beam against the surviving shapes.  No LDS, no atomics.
//   * localize  : one wave per vehicle; candidate lanes come from the static grid cell under the
//                 vehicle; the convex-hull containment test spreads hull edges over the 64 lanes and
//                 votes with a ballot.
//   * contacts  : one wave per vehicle; lanes = the other movers (SAT each), then lanes = quads of
//                 the grid cells under the chassis AABB; flag words are OR-combined by ballots.
//   * trigger   : wavefront min-reduce (DPP shuffles) over the pending traffic blocks.
//   * integrate / observe / IDM : one thread per entity over include/md_entity.h.
//
// There is no dense contraction anywhere on this path, hence no MFMA.  The arithmetic formulas are
// the shared include/md_geom.h / md_entity.h ones (bit-exact vs the CPU oracle, -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "md_scenario.h"

namespace {

constexpr int kBlock = 256;  // block size of the auxiliary kernels; env_kernel is templated on its own

enum Phase : int {
    PH_RESET = 1,
    PH_IDM = 2,
    PH_INTEGRATE = 4,
    PH_LOCALIZE = 8,
    PH_CONTACTS = 16,
    PH_TRAFFIC = 32,
    PH_OBSERVE = 64,
    PH_LIDAR = 128,
    PH_LIFECYCLE = 256,
    PH_ALL = 511,
};

thread_local char g_err[256] = "ok";

// In-kernel phase stamps: DIAGNOSTIC BUILD ONLY (-DMD_STAMP, tools/stamp_profile.py).  The stamp
// values go to a buffer of their own that no kernel code reads; the production build contains none.
#ifdef MD_STAMP
__device__ unsigned long long* g_stamp_buf = nullptr;
__device__ const int* g_env_order = nullptr;
// slots 0..11: shader-cycle stamps; 12 / 13: s_memrealtime (100 MHz, one time base for the whole chip) at the first / last
// stamp; 14: HW_ID | XCC_ID << 32 (which CU the workgroup ran on) -- tools/timeline_probe.py rebuilds the occupancy timeline
#define MD_STAMP_AT(i)                                                                         \
    do {                                                                                       \
        if (threadIdx.x == 0 && g_stamp_buf) {                                                 \
            unsigned long long* sb_ = g_stamp_buf + (size_t)blockIdx.x * 32;                   \
            sb_[(i)] = __builtin_readcyclecounter();                                           \
            if ((i) == 0) {                                                                    \
                sb_[12] = __builtin_amdgcn_s_memrealtime();                                    \
                sb_[14] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |               \
                          ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);        \
            }                                                                                  \
            if ((i) == 11) sb_[13] = __builtin_amdgcn_s_memrealtime();                         \
        }                                                                                      \
    } while (0)
#define MD_FINE_STAMP(cond, i)                                                                 \
    do {                                                                                       \
        if ((cond) && g_stamp_buf) g_stamp_buf[(size_t)blockIdx.x * 32 + 16 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define MD_STAMP_AT(i) do { } while (0)
#define MD_FINE_STAMP(cond, i) do { } while (0)
#endif
// Diagnostic builds only (tools/ab/env_knockout.sh): stages of the fused env kernel left out to read their marginal cost off the
// launch time.  0 in the product.
#ifndef MD_ENV_SKIP
#define MD_ENV_SKIP 0
#endif
// the single-agent kernels' lidar sectors by ticket as well (the waves that idle beside the observation / IDM stage take them all)
#ifndef MD_LEAN_LIDAR_TICKETS
#define MD_LEAN_LIDAR_TICKETS 1
#endif

__device__ __forceinline__ float bcast_f(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int bcast_i(int v, int src) { return __shfl(v, src, 64); }

// The env state is read once and written once per step: streaming accesses (the nt hint keeps them from displacing the map
// tables -- read again next step by the same env, on the same XCD -- from L2).  Measured: workgroup kernel 88.6 -> 86.7 us,
// wave kernel on 4096 distinct maps 93.5 -> 89.0 us (-DMD_NT_STAGE=0 switches the hint off).
#ifndef MD_NT_STAGE
#define MD_NT_STAGE 1
#endif
typedef unsigned int md_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream(const uint4* p) {
#if MD_NT_STAGE
    const md_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const md_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_stream(uint4* p, const uint4 v) {
#if MD_NT_STAGE
    md_u32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<md_u32x4*>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_stream_f(float* p, float v) {
#if MD_NT_STAGE
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// Wavefront min-reduce over the lanes whose `valid` is set; returns `none` when no lane is valid.
// Built from a 64-bit ballot + v_readlane walks over the set bits: on gfx950 __shfl_xor lowers to
// ds_bpermute (an LDS-crossbar round trip of ~100+ cycles per step), and the candidate sets here are
// sparse (a handful of lanes), so walking the ballot is several times faster than a 6-step butterfly.
__device__ __forceinline__ int wave_min_i(int v, bool valid, int none) {
    unsigned long long m = __ballot(valid);
    int best = none;
    while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int o = __shfl(v, l, 64);
        best = o < best ? o : best;
    }
    return best;
}

// ------------------------------------------------------------------------------------------------
// Lidar for one (agent, sector) work item, executed by one wave.
// ------------------------------------------------------------------------------------------------
// det: LDS words [2] of agent a's detected set (nullptr = not tracked)
__device__ __forceinline__ void lidar_item(const MdWorld& w, const MdState& s, const MdConfig& c, int a, int sec, int lane,
                           float* __restrict__ out_row, unsigned long long* det) {
    const MdShape me = s.shape[a];  // wave-uniform (s is the env-local, LDS-staged view)
    const int beam = sec * 64 + lane;
    const bool valid = beam < c.n_beams;
    if (!md_present(me.flags)) {
        if (valid) out_row[beam] = 1.0f;
        return;
    }
    float bc = 1.0f, bs = 0.0f;
    if (valid) {
        bc = w.beam_cs[2 * beam];
        bs = w.beam_cs[2 * beam + 1];
    }
    const float dirx = (bc * me.c - bs * me.s) * c.lidar_range;
    const float diry = (bs * me.c + bc * me.s) * c.lidar_range;

    // Sector cone for culling: axis = middle of the sector's beam fan, half-span h.  The cull is only ever
    // CONSERVATIVE (2e-3 slack on the cosine, 1 cm on the radii), so it runs on the hardware's approximate
    // rcp / rsq / sqrt / sin / cos (1 ulp, v_sin/v_cos take revolutions); the cast below stays exact.
    const int nb = min(64, c.n_beams - sec * 64);
    const float inv_n = __builtin_amdgcn_rcpf((float)c.n_beams);
    const float mid_rev = ((float)(sec * 64) + 0.5f * (float)(nb - 1)) * inv_n;   // axis angle, in revolutions, ego frame
    const float half_rev = 0.5f * (float)(nb - 1) * inv_n;
    const float ax = __builtin_amdgcn_cosf(mid_rev), ay = __builtin_amdgcn_sinf(mid_rev);
    const float cos_h = __builtin_amdgcn_cosf(half_rev), sin_h = __builtin_amdgcn_sinf(half_rev);
    const float half_span = MD_TWO_PI_F * half_rev;
    // cone test is meaningful only while h + asin(rb/dist) < pi; asin <= pi/2, so narrow sectors never need the check
    const bool narrow = half_span < MD_HALF_PI_F - 0.02f;
    const float wax = ax * me.c - ay * me.s;  // axis in world frame
    const float way = ay * me.c + ax * me.s;

    float best = 1.0f;
    int best_j = -1;  // slot of the body this beam hits first (tracked only when det != nullptr)
    for (int j0 = 0; j0 < c.cap; j0 += 64) {
        const int j = j0 + lane;
        MdShape o;
        o.flags = 0;
        if (j < c.cap) o = s.shape[j];
        bool keep = (j < c.cap) && (j != a) && md_present(o.flags);
        if (keep) {
            const float ddx = o.cx - me.cx, ddy = o.cy - me.cy;
            const float d2 = ddx * ddx + ddy * ddy;
            const int k = md_kind_of(o.flags);
            const float rb = (md_is_circle_kind(k) ? o.hl : __builtin_amdgcn_sqrtf(o.hl * o.hl + o.hw * o.hw)) + 0.01f;
            const float reach = c.lidar_range + rb;
            if (d2 > reach * reach * 1.0001f) keep = false;
            else if (d2 > rb * rb) {
                // cone test: angle(d, axis) <= h + alpha, sin(alpha) = rb / dist
                const float inv = __builtin_amdgcn_rsqf(d2);
                const float sin_a = rb * inv;
                const float cos_a = __builtin_amdgcn_sqrtf(md_max(0.0f, 1.0f - sin_a * sin_a));
                if (narrow || half_span + md_asin(md_min(sin_a, 1.0f)) < MD_PI_F - 0.01f) {
                    const float cos_lim = cos_h * cos_a - sin_h * sin_a;  // cos(h + alpha)
                    const float cos_t = (ddx * wax + ddy * way) * inv;
                    if (cos_t < cos_lim - 2e-3f) keep = false;
                }
            }
        }
        unsigned long long mask = __ballot(keep);
        while (mask) {
            const int k = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const float ocx = bcast_f(o.cx, k), ocy = bcast_f(o.cy, k);
            const float oc = bcast_f(o.c, k), os = bcast_f(o.s, k);
            const float ohl = bcast_f(o.hl, k), ohw = bcast_f(o.hw, k);
            const int ofl = bcast_i(o.flags, k);
            const float t = md_ray_shape(me.cx, me.cy, dirx, diry, ocx, ocy, oc, os, ohl, ohw, md_kind_of(ofl));
            if (t < best) {
                best = t;
                best_j = j0 + k;  // candidates come in ascending slot order: equal fractions keep the lowest slot
            }
        }
    }
    if (valid) st_stream_f(&out_row[beam], best);
    if (det) {
        // union of the 64 beams' first hits: peel one distinct slot per iteration (<= a handful)
        unsigned long long todo = __ballot(valid && best_j >= 0);
        unsigned long long lo = 0ull, hi = 0ull;
        while (todo) {
            const int k = __ffsll((long long)todo) - 1;
            const int jj = bcast_i(best_j, k);
            todo &= ~__ballot(best_j == jj);
            if (jj < 64) lo |= 1ull << jj;
            else hi |= 1ull << (jj - 64);
        }
        if (lane == 0) {
            if (lo) atomicOr(&det[0], lo);
            if (hi) atomicOr(&det[1], hi);
        }
    }
}

// (Keeping the sectors off the wave that runs the agent's observe chain was tried: 154 vs 145 us, worse.)
// `tickets` (LDS counter, zeroed by the caller behind a barrier) = the waves take the (agent, sector) items as they get free -- the
// multi-agent kernels, whose workgroups are all resident at once so that a launch lasts as long as its slowest env; nullptr = dealt
// round-robin.  The items are independent: the order changes nothing.
__device__ __forceinline__ void phase_lidar(const MdWorld& w, const MdState& s, const MdConfig& c, int e, int tid, int kWaves,
                            float* out, int out_stride, int out_offset, unsigned long long* l_det, int* tickets = nullptr) {  // `out` is the GLOBAL output base
    const int wave = tid >> 6, lane = tid & 63;
    const int nsec = (c.n_beams + 63) >> 6;
    const int items = c.agents_per_env * nsec;
    if (tickets) {
        for (int guard = 0; guard < items; ++guard) {
            int it = 0;
            if (lane == 0) it = atomicAdd(tickets, 1);
            it = __builtin_amdgcn_readfirstlane(it);
            if (it < 0 || it >= items) break;
            const int a = it / nsec, sec = it - a * nsec;
            float* row = out + (size_t)(e * c.agents_per_env + a) * out_stride + out_offset;
            lidar_item(w, s, c, a, sec, lane, row, l_det ? l_det + 2 * a : nullptr);
        }
        return;
    }
    for (int it = wave; it < items; it += kWaves) {
        const int a = it / nsec, sec = it - a * nsec;
        float* row = out + (size_t)(e * c.agents_per_env + a) * out_stride + out_offset;
        lidar_item(w, s, c, a, sec, lane, row, l_det ? l_det + 2 * a : nullptr);
    }
}

// The detectors' cull record of quad q: centre, radius of a circle around it, kind -- from MdWorld.quad_ball (16 bytes) or,
// without that table, from the quad itself.
struct QuadBall { float mx, my, rr; int kind; };
__device__ __forceinline__ QuadBall quad_ball_of(const MdWorld& w, int q) {
    QuadBall b;
    if (w.quad_ball) {
        const float4 t = reinterpret_cast<const float4*>(w.quad_ball)[q];
        b.mx = t.x;
        b.my = t.y;
        b.rr = t.z;
        b.kind = __float_as_int(t.w);
    } else {
        const float4* quads4 = reinterpret_cast<const float4*>(w.quads);
        const float4 lo = quads4[2 * (size_t)q], hi = quads4[2 * (size_t)q + 1];
        b.mx = 0.25f * (lo.x + lo.z + hi.x + hi.z);
        b.my = 0.25f * (lo.y + lo.w + hi.y + hi.w);
        const float r2 = md_max(md_max((lo.x - b.mx) * (lo.x - b.mx) + (lo.y - b.my) * (lo.y - b.my),
                                       (lo.z - b.mx) * (lo.z - b.mx) + (lo.w - b.my) * (lo.w - b.my)),
                                md_max((hi.x - b.mx) * (hi.x - b.mx) + (hi.y - b.my) * (hi.y - b.my),
                                       (hi.z - b.mx) * (hi.z - b.mx) + (hi.w - b.my) * (hi.w - b.my)));
        b.rr = md_sqrt(r2) * 1.01f + 1.0e-3f;
        b.kind = w.quad_kind[q];
    }
    return b;
}

// One detector fan of one agent against the quads [qa, qb) of its map by ONE wave, in two phases so that the expensive
// slab tests run on full wavefronts: (1) lanes = quads: kind, reach, then every beam against the quad's bounding circle
// (a dozen instructions); the (quad, beam) pairs that survive -- a few per hundred -- are appended to `pairs` (LDS,
// kDetPairs ints of this wave) through ballot prefix counts; (2) lanes = pairs: fetch the quad, cast, atomicMin on the
// fraction's bit pattern in `best` (initialised to 1.0).  The list is drained whenever a pass could overflow it.  Same
// arithmetic and the same minima as line_detector_kernel and the oracle's serial loop.
constexpr int kDetPairs = 512;
__device__ __forceinline__ void detector_drain(const MdWorld& w, const MdShape& me, const float* beam_cs, float range, const int* pairs, int n,
                               int qa, int* best, int lane_id) {   // a pair = (quad - qa) << 8 | beam
    const float4* quads4 = reinterpret_cast<const float4*>(w.quads);
    for (int k = lane_id; k < n; k += 64) {
        const int pr = pairs[k];
        const int q = qa + (pr >> 8), i = pr & 255;
        const float4 lo = quads4[2 * (size_t)q], hi = quads4[2 * (size_t)q + 1];
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const float bc = beam_cs[2 * i], bs = beam_cs[2 * i + 1];
        const float ux = bc * me.c - bs * me.s, uy = bs * me.c + bc * me.s;
        const float t = md_ray_quad(me.cx, me.cy, ux * range, uy * range, v);
        if (t < 1.0f) atomicMin(&best[i], __float_as_int(t));
    }
}

__device__ __forceinline__ void detector_wave(const MdWorld& w, const MdShape& me, int qa, int qb, const float* beam_cs, int n_beams,
                              float range, uint32_t kind_mask, int* best, int* pairs, int lane_id) {
    const float reach = range * 1.001f;
    // is the table a uniform fan?  beam i = (cos, sin)(phase0 + i 2 pi / n): one lane per beam compares
    const float phase0 = atan2f(beam_cs[1], beam_cs[0]);
    const float dphi = 6.283185307179586f / (float)n_beams, inv_dphi = 1.0f / dphi;
    bool fan_ok = true;
    for (int i = lane_id; i < n_beams; i += 64) {
        float sn, cs;
        sincosf(phase0 + (float)i * dphi, &sn, &cs);
        fan_ok = fan_ok && md_fabs(cs - beam_cs[2 * i]) < 1.0e-3f && md_fabs(sn - beam_cs[2 * i + 1]) < 1.0e-3f;
    }
    const bool uniform_fan = __ballot(!fan_ok) == 0ull && n_beams >= 4;
    int cnt = 0;   // wave-uniform
    // the 16-byte records of the NEXT 64 quads are requested before this round's are worked on: the rounds are a dependent chain
    // (ballot, pair list), and a cold fetch per round was most of the phase
    QuadBall nb;
    nb.mx = nb.my = nb.rr = 0.0f;
    nb.kind = 0;
    if (qa + lane_id < qb) nb = quad_ball_of(w, qa + lane_id);
    for (int q0 = qa; q0 < qb; q0 += 64) {
        const int q = q0 + lane_id;
        bool near = false;
        float px = 0.0f, py = 0.0f, rr = 0.0f;
        const QuadBall b = nb;
        if (q + 64 < qb) nb = quad_ball_of(w, q + 64);
        if (q < qb) {
            px = b.mx - me.cx;
            py = b.my - me.cy;
            rr = b.rr;
            const float far = reach + rr;
            near = ((kind_mask >> b.kind) & 1u) && !(px * px + py * py > far * far);
        }
        if (__ballot(near) == 0ull) continue;
        // Which beams can meet this quad at all?  For a uniform fan (beam i at phase0 + i dphi from the heading -- every
        // detector table is one; checked below, else all beams are tried) only those within asin(r / d) of the direction
        // to the quad's centre: a handful instead of all 12 ... 72.  Hardware atan2 / asin are good enough here, a margin
        // covers them and the exact circle test follows anyway.
        int i_lo = 0, n_try = n_beams;
        if (uniform_fan && near) {
            const float d2 = px * px + py * py;
            if (d2 > rr * rr * 1.0201f) {
                const float lx = px * me.c + py * me.s, ly = py * me.c - px * me.s;   // the centre in the agent's frame
                const float half = asinf(md_min(rr * 1.01f * rsqrtf(d2), 1.0f)) + 0.02f;
                float rel = atan2f(ly, lx) - phase0;
                const float lo = (rel - half) * inv_dphi, hi = (rel + half) * inv_dphi;
                i_lo = (int)floorf(lo);
                n_try = min((int)ceilf(hi) - i_lo + 1, n_beams);
                i_lo = ((i_lo % n_beams) + n_beams) % n_beams;
            }
        }
        int max_try = near ? n_try : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) max_try = max(max_try, __shfl_xor(max_try, off, 64));
        for (int k = 0; k < max_try; ++k) {
            int i = i_lo + k;
            if (i >= n_beams) i -= n_beams;
            const bool mine_ = near && k < n_try;
            const float bc = beam_cs[2 * i], bs = beam_cs[2 * i + 1];
            const float ux = bc * me.c - bs * me.s, uy = bs * me.c + bc * me.s;
            const float perp = ux * py - uy * px, along = ux * px + uy * py;
            const bool pass = mine_ && !(md_fabs(perp) > rr * 1.001f + 1.0e-3f || along < -rr || along > reach + rr);
            const unsigned long long m = __ballot(pass);
            if (m == 0ull) continue;
            if (cnt + 64 > kDetPairs) {   // keep room for a whole ballot
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                detector_drain(w, me, beam_cs, range, pairs, cnt, qa, best, lane_id);
                __builtin_amdgcn_wave_barrier();
                cnt = 0;
            }
            if (pass) pairs[cnt + __popcll(m & ((1ull << lane_id) - 1ull))] = ((q - qa) << 8) | i;
            cnt += __popcll(m);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    detector_drain(w, me, beam_cs, range, pairs, cnt, qa, best, lane_id);
    __builtin_amdgcn_wave_barrier();
}

// One or TWO detector fans of one agent (side detector + lane-line detector: different beam tables, ranges and line kinds) against
// the quads [qa, qb) of its map by ONE wave, in ONE pass over the quads' cull records, in two phases so that the expensive slab
// tests run on full wavefronts: (1) lanes = quads: per fan kind, reach, then the beams that can meet the quad's bounding circle (a
// dozen instructions each); the (quad, beam) pairs that survive -- a few per hundred -- are appended to `pairs` (LDS, kDetPairs
// ints of this wave) through ballot prefix counts, the second fan's beams numbered behind the first's; (2) lanes = pairs: fetch
// the quad, cast, atomicMin on the fraction's bit pattern in the fan's `best` (initialised to 1.0).  The list is drained whenever
// a pass could overflow it.  Same arithmetic and the same minima as the oracle's serial loop per detector.
constexpr int kAllBeamsMax = 8;
struct DetFan2 {
    const float* cs0; int n0; float range0; uint32_t mask0; int* best0;
    const float* cs1; int n1; float range1; uint32_t mask1; int* best1;   // n1 == 0: one fan only
};
__device__ __forceinline__ void detector_drain2(const MdWorld& w, const MdShape& me, const DetFan2& f, const int* pairs, int n, int qa,
                               int lane_id) {   // a pair = (quad - qa) << 8 | beam (fan 1's beams from n0 on)
    const float4* quads4 = reinterpret_cast<const float4*>(w.quads);
    for (int k = lane_id; k < n; k += 64) {
        const int pr = pairs[k];
        const int q = qa + (pr >> 8), ig = pr & 255;
        const bool second = ig >= f.n0;
        const int i = second ? ig - f.n0 : ig;
        const float* cs = second ? f.cs1 : f.cs0;
        const float range = second ? f.range1 : f.range0;
        const float4 lo = quads4[2 * (size_t)q], hi = quads4[2 * (size_t)q + 1];
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const float bc = cs[2 * i], bs = cs[2 * i + 1];
        const float ux = bc * me.c - bs * me.s, uy = bs * me.c + bc * me.s;
        const float t = md_ray_quad(me.cx, me.cy, ux * range, uy * range, v);
        if (t < 1.0f) atomicMin(second ? &f.best1[i] : &f.best0[i], __float_as_int(t));
    }
}

// is the table a uniform fan?  beam i = (cos, sin)(phase0 + i 2 pi / n): one lane per beam compares
__device__ __forceinline__ bool detector_uniform_fan(const float* beam_cs, int n_beams, int lane_id, float* phase0_out) {
    const float phase0 = atan2f(beam_cs[1], beam_cs[0]);
    const float dphi = 6.283185307179586f / (float)n_beams;
    bool fan_ok = true;
    for (int i = lane_id; i < n_beams; i += 64) {
        float sn, cs;
        sincosf(phase0 + (float)i * dphi, &sn, &cs);
        fan_ok = fan_ok && md_fabs(cs - beam_cs[2 * i]) < 1.0e-3f && md_fabs(sn - beam_cs[2 * i + 1]) < 1.0e-3f;
    }
    *phase0_out = phase0;
    return __ballot(!fan_ok) == 0ull && n_beams >= 4;
}

__device__ __forceinline__ void detector_wave2(const MdWorld& w, const MdShape& me, int qa, int qb, const DetFan2& f, int* pairs, int lane_id) {
    float ph0 = 0.0f, ph1 = 0.0f;
    const bool uni0 = detector_uniform_fan(f.cs0, f.n0, lane_id, &ph0);
    const bool uni1 = f.n1 > 0 ? detector_uniform_fan(f.cs1, f.n1, lane_id, &ph1) : false;
    const float reach_max = md_max(f.range0, f.n1 > 0 ? f.range1 : 0.0f) * 1.001f;
    int cnt = 0;   // wave-uniform
    // the 16-byte records of the NEXT 64 quads are requested before this round's are worked on: the rounds are a dependent chain
    // (ballot, pair list), and a cold fetch per round was most of the phase
    QuadBall nb;
    nb.mx = nb.my = nb.rr = 0.0f;
    nb.kind = 0;
    if (qa + lane_id < qb) nb = quad_ball_of(w, qa + lane_id);
    // Two passes: (a) lanes = quads, reach only, the quads within reach COMPACTED onto a list (the first half of `pairs`); (b) lanes =
    // listed quads, dense: kinds, beam ranges, circle tests.  In one pass (b)'s work ran for the whole wave as soon as one of a
    // round's 64 quads was within reach, and the quads within reach (a third of a map's) are spread over most rounds.
    constexpr int kNear = kDetPairs / 2, kPairs = kDetPairs - kNear;
    int* near_list = pairs;            // [kNear]  quad - qa
    int* pair_list = pairs + kNear;    // [kPairs]
    int n_near = 0;                    // wave-uniform
    const bool across_rounds = f.n0 > kAllBeamsMax || f.n1 > kAllBeamsMax;
    auto work_off = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int t0 = 0; t0 < n_near; t0 += 64) {
        const bool in_reach = t0 + lane_id < n_near;
        const int q0 = qa + (in_reach ? near_list[t0 + lane_id] : 0);   // this lane's quad (no longer q0 + lane)
        float px = 0.0f, py = 0.0f, rr = 0.0f;
        QuadBall b;
        b.mx = b.my = b.rr = 0.0f;
        b.kind = 0;
        if (in_reach) {
            b = quad_ball_of(w, q0);
            px = b.mx - me.cx;
            py = b.my - me.cy;
            rr = b.rr;
        }
        // one fan after the other against this round's quads (the second only where there is one)
        for (int fan = 0; fan < (f.n1 > 0 ? 2 : 1); ++fan) {
            const float* beam_cs = fan ? f.cs1 : f.cs0;
            const int n_beams = fan ? f.n1 : f.n0;
            const float reach = (fan ? f.range1 : f.range0) * 1.001f;
            const uint32_t kind_mask = fan ? f.mask1 : f.mask0;
            const float phase0 = fan ? ph1 : ph0;
            const bool uniform_fan = fan ? uni1 : uni0;
            const int beam_base = fan ? f.n0 : 0;
            const float inv_dphi = 1.0f / (6.283185307179586f / (float)n_beams);
            const float far = reach + rr;
            const bool near = in_reach && ((kind_mask >> b.kind) & 1u) && !(px * px + py * py > far * far);
            if (__ballot(near) == 0ull) continue;
            // Which beams can meet this quad at all?  For a uniform fan (beam i at phase0 + i dphi from the heading -- every
            // detector table is one; checked above, else all beams are tried) only those within asin(r / d) of the direction
            // to the quad's centre: a handful instead of all 12 ... 72.  Hardware atan2 / asin are good enough here, a margin
            // covers them and the exact circle test follows anyway.
            int i_lo = 0, n_try = n_beams;
            if (uniform_fan && near && n_beams > kAllBeamsMax) {   // a handful of beams: trying all costs less than asin + atan2
                const float d2 = px * px + py * py;
                if (d2 > rr * rr * 1.0201f) {
                    const float lx = px * me.c + py * me.s, ly = py * me.c - px * me.s;   // the centre in the agent's frame
                    const float half = asinf(md_min(rr * 1.01f * rsqrtf(d2), 1.0f)) + 0.02f;
                    float rel = atan2f(ly, lx) - phase0;
                    const float lo = (rel - half) * inv_dphi, hi = (rel + half) * inv_dphi;
                    i_lo = (int)floorf(lo);
                    n_try = min((int)ceilf(hi) - i_lo + 1, n_beams);
                    i_lo = ((i_lo % n_beams) + n_beams) % n_beams;
                }
            }
            // The (quad, beam) candidates of this round, FLAT: lane t of a pass takes candidate t -- beam k of the quad in the lane l
            // that owns it (the largest l whose exclusive prefix count is <= t: a six-step search through the wave's registers).
            // A loop over k as long as the NEAREST quad needs (up to all 72 beams, one lane busy) ran 5x as many passes.
            const int nt = near ? n_try : 0;
            int pin = nt;   // inclusive prefix count
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(pin, off, 64);
                if (lane_id >= off) pin += v;
            }
            const int total = __shfl(pin, 63, 64);   // wave-uniform
            const int pex = pin - nt;
            for (int t0 = 0; t0 < total; t0 += 64) {
                const int t = t0 + lane_id;
                int l = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const int cnd = l + step;
                    const int pc = __shfl(pex, cnd & 63, 64);
                    if (cnd < 64 && pc <= t) l = cnd;
                }
                const bool mine_ = t < total;
                const int k = t - __shfl(pex, l, 64);
                const float px_ = __shfl(px, l, 64), py_ = __shfl(py, l, 64), rr_ = __shfl(rr, l, 64);
                int i = __shfl(i_lo, l, 64) + k;
                if (i >= n_beams) i -= n_beams;
                if (!mine_) i = 0;
                const int q_ = __shfl(q0, l, 64);
                const float bc = beam_cs[2 * i], bs = beam_cs[2 * i + 1];
                const float ux = bc * me.c - bs * me.s, uy = bs * me.c + bc * me.s;
                const float perp = ux * py_ - uy * px_, along = ux * px_ + uy * py_;
                const bool pass = mine_ && !(md_fabs(perp) > rr_ * 1.001f + 1.0e-3f || along < -rr_ || along > reach + rr_);
                const unsigned long long m = __ballot(pass);
                if (m == 0ull) continue;
                if (cnt + 64 > kPairs) {   // keep room for a whole ballot
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    detector_drain2(w, me, f, pair_list, cnt, qa, lane_id);
                    __builtin_amdgcn_wave_barrier();
                    cnt = 0;
                }
                if (pass) pair_list[cnt + __popcll(m & ((1ull << lane_id) - 1ull))] = ((q_ - qa) << 8) | (beam_base + i);
                cnt += __popcll(m);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();   // the list is free again
    n_near = 0;
    };
    for (int q0 = qa; q0 < qb; q0 += 64) {
        const int q = q0 + lane_id;
        const QuadBall b = nb;
        if (q + 64 < qb) nb = quad_ball_of(w, q + 64);
        bool in_reach = false;
        if (q < qb) {
            const float px = b.mx - me.cx, py = b.my - me.cy;
            const float far = reach_max + b.rr;
            in_reach = !(px * px + py * py > far * far);
        }
        const unsigned long long m = __ballot(in_reach);
        if (m == 0ull) continue;
        if (n_near + 64 > kNear) work_off();
        if (in_reach) near_list[n_near + __popcll(m & ((1ull << lane_id) - 1ull))] = q - qa;
        n_near += __popcll(m);
        if (!across_rounds) work_off();   // fans of a few beams: round by round (measured: collecting across rounds costs them 1 %)
    }
    work_off();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    detector_drain2(w, me, f, pair_list, cnt, qa, lane_id);
    __builtin_amdgcn_wave_barrier();
}

// Side / lane-line detector as an entry point of its own (md_line_detector).  Work items = (agent, part of the map's quads),
// one WAVE each running detector_wave above; with four or more agents per env an item is a whole agent, with fewer the quads
// are split four ways.  A workgroup takes FOUR consecutive items of one env (grid = n_envs x ceil(items / 4)): the 40-agent
// tollgate batch of 512 envs is 5 120 workgroups -- the first form of this launch, one workgroup per env walking its ten
// rounds of agents in turn, left the chip at two workgroups per CU (174 us per launch there).  One thread per beam walking
// all quads, the very first form, took 1 ms on a scenario scene's ~1100 line pieces.
__device__ __host__ inline int detector_parts(int A) { return (A >= kBlock / 64) ? 1 : kBlock / 64; }
__device__ __host__ inline int detector_groups(int A) { return (A * detector_parts(A) + kBlock / 64 - 1) / (kBlock / 64); }

__global__ __launch_bounds__(kBlock) void line_detector_kernel(MdWorld w, MdState s, MdConfig c,
                                                               const float* __restrict__ beam_cs, int n_beams,
                                                               float range, uint32_t kind_mask, float* out,
                                                               int out_stride, int out_offset,
                                                               const float* __restrict__ beam_cs1, int n_beams1, float range1,
                                                               uint32_t kind_mask1, int out_offset1) {
    // (beam_cs1, n_beams1 > 0, ...): a SECOND fan evaluated in the same pass over the quads (md_line_detectors)
    extern __shared__ int l_ld[];
    constexpr int kW = kBlock / 64;
    const int A = c.agents_per_env;
    const int nb = n_beams + n_beams1;
    const int parts = detector_parts(A), groups = detector_groups(A);
    int* l_best = l_ld;                                              // [kW * nb] bit patterns of the closest fractions
    float* l_bm = reinterpret_cast<float*>(l_ld + kW * nb);          // [nb][2] the beam tables, fan 0 then fan 1
    int* l_pairs = reinterpret_cast<int*>(l_bm + 2 * nb);            // [kW][kDetPairs]
    const int e = blockIdx.x / groups, grp = blockIdx.x - e * groups;
    if (e >= c.n_envs) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = w.env_map[e];
    const int q0 = w.quad_off[m], q1 = w.quad_off[m + 1];
    const int a_first = (grp * kW) / parts;                          // first agent this workgroup serves
    const int n_here = min(A - a_first, kW / parts);                 // agents it serves (4, or 1 when the quads are split)
    for (int it = tid; it < n_here * nb; it += kBlock) l_best[it] = __float_as_int(1.0f);
    for (int it = tid; it < 2 * n_beams; it += kBlock) l_bm[it] = beam_cs[it];
    for (int it = tid; it < 2 * n_beams1; it += kBlock) l_bm[2 * n_beams + it] = beam_cs1[it];
    __syncthreads();
    const int per = (q1 - q0 + parts - 1) / parts;
    const int it = grp * kW + wave;
    if (it < A * parts) {
        const int a = it / parts, part = it - a * parts;
        const MdShape me = s.shape[e * c.cap + a];
        if (md_present(me.flags)) {
            const int qa = q0 + part * per, qb = min(qa + per, q1);
            DetFan2 f;
            f.cs0 = l_bm; f.n0 = n_beams; f.range0 = range; f.mask0 = kind_mask; f.best0 = l_best + (a - a_first) * nb;
            f.cs1 = l_bm + 2 * n_beams; f.n1 = n_beams1; f.range1 = range1; f.mask1 = kind_mask1; f.best1 = f.best0 + n_beams;
            detector_wave2(w, me, qa, qb, f, l_pairs + wave * kDetPairs, lane);
        }
    }
    __syncthreads();
    for (int k = tid; k < n_here * nb; k += kBlock) {
        const int al = k / nb, i = k - al * nb;
        float* row = out + (size_t)(e * A + a_first + al) * out_stride;
        if (i < n_beams) row[out_offset + i] = __int_as_float(l_best[k]);
        else row[out_offset1 + (i - n_beams)] = __int_as_float(l_best[k]);
    }
}

// ------------------------------------------------------------------------------------------------
// Grid helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int grid_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ------------------------------------------------------------------------------------------------
// Localisation, one wave per vehicle.
// ------------------------------------------------------------------------------------------------
// onlane_out: nullptr = write the ON_LANE bit into s.flags[n] (stand-alone phase); otherwise store the bare
// decision there and leave s.flags alone (fused step: contacts run concurrently and own the other bits).
__device__ __forceinline__ void localize_vehicle(const MdWorld& w, const MdLane* lanes, const MdRoad* roads, const MdState& s, int e,
                                 int n, int lane_id, uint32_t* onlane_out = nullptr) {
    // n = slot inside the env-local view; lanes / roads = this env's map tables (LDS copies)
    const MdShape sh = s.shape[n];
    if (!md_drives(sh.flags)) return;
    MdNav nav = s.nav[n];
    const int m = w.env_map[e];
    const int32_t* rroads = s.route_roads + (size_t)n * MD_ROUTE_LEN;   // global: read only when the cursors advance
    const int32_t* rnodes = s.route_nodes + (size_t)n * MD_ROUTE_LEN;
    const int cur_road = nav.road0;
    const bool has_next = nav.ck1 != nav.ck0;
    const int next_road = has_next ? nav.road1 : -1;

    MD_FINE_STAMP(n == 0 && lane_id == 0, 0);
    const MdGrid g = w.grid[m];
    const int gx = (int)md_floor((sh.cx - g.x0) * g.inv_cell);
    const int gy = (int)md_floor((sh.cy - g.y0) * g.inv_cell);
    int it0 = 0, it1 = 0;
    if (gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny) {
        const int cell = g.cell_base + gy * g.nx + gx;
        it0 = w.cell_start[cell];
        it1 = w.cell_start[cell + 1];
    }
    MD_FINE_STAMP(n == 0 && lane_id == 0 && it1 >= 0, 1);
    int on_lane = 0;
    int best_any = -1, best_cur = -1, best_next = -1;
    float d_any = 3.0e38f, d_cur = 3.0e38f, d_next = 3.0e38f;
    float ls_any = 0.0f, ls_cur = 0.0f, ls_next = 0.0f;
    // Candidate lanes of the cell: each LANE fetches one cell item and tests its lane record's hull
    // AABB (records are in LDS); survivors are visited in ascending lane id through the ballot mask
    // (ties in distance resolve to the lowest lane id, like the oracle's ascending scan).
    //   pass A  hull containment of every surviving candidate (edges spread over the 64 lanes, ballot
    //           vote; 4-vertex hulls come inline from the lane record, no second lookup)
    //   pass B  ONE batched Frenet evaluation: lane k evaluates the k-th contained candidate
    //   pass C  uniform selection over the (few) contained candidates
    for (int itb = it0; itb < it1; itb += 64) {
        const int it = itb + lane_id;
        int l = -1;
        bool pass = false;
        if (it < it1) {
            l = w.cell_items[it];
            if (l >= 0) {
                const MdLane* L = &lanes[l];
                pass = !(sh.cx < L->x0 || sh.cx > L->x1 || sh.cy < L->y0 || sh.cy > L->y1);
            }
        }
        unsigned long long mask = __ballot(pass);
        MD_FINE_STAMP(n == 0 && lane_id == 0, 2);
#ifdef MD_STAMP
        if (n == 0 && lane_id == 0 && g_stamp_buf)
            g_stamp_buf[(size_t)blockIdx.x * 32 + 16 + 15] = (unsigned long long)__popcll(mask) | ((unsigned long long)(it1 - it0) << 32);
#endif
        unsigned long long inside = 0ull;  // bit k: candidate held by lane k contains the point
        while (mask) {
            const int k = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const MdLane* L = &lanes[bcast_i(l, k)];
            const int hn = L->hull_n;
            const float* xy = md_lane_hull(L, w.hull_xy);
            bool outside = false;
            for (int i = lane_id; i < hn; i += 64) {
                const int j = (i + 1 == hn) ? 0 : i + 1;
                const float ex = xy[2 * j] - xy[2 * i], ey = xy[2 * j + 1] - xy[2 * i + 1];
                const float cr = ex * (sh.cy - xy[2 * i + 1]) - ey * (sh.cx - xy[2 * i]);
                if (cr < 0.0f) outside = true;
            }
            if (__ballot(outside) == 0ull && hn >= 3) inside |= 1ull << k;
        }
        if (inside == 0ull) continue;
        on_lane = 1;
        // pass B: my lane's candidate (if contained): Frenet coordinates, heading filter, L1 distance
        float my_dist = 3.0e38f, my_ls = 0.0f;
        int my_road = -1;
        if ((inside >> lane_id) & 1ull) {
            const MdLane* L = &lanes[l];
            float llat;
            md_lane_local(L, sh.cx, sh.cy, &my_ls, &llat);
            if (md_lane_heading_dot(L, sh.cx, sh.cy, sh.c, sh.s) > 0.0f) {
                my_dist = md_lane_distance(L, my_ls, llat);
                my_road = L->road;
            }
        }
        // pass C (the winner's longitudinal is kept: the checkpoint update below needs it)
        while (inside) {
            const int k = __ffsll((long long)inside) - 1;
            inside &= inside - 1;
            const float dist = bcast_f(my_dist, k);
            const int road = bcast_i(my_road, k);
            const int lk = bcast_i(l, k);
            if (road < 0) continue;  // failed the heading filter
            const float lsk = bcast_f(my_ls, k);
            if (dist < d_any) { d_any = dist; best_any = lk; ls_any = lsk; }
            if (road == cur_road && dist < d_cur) { d_cur = dist; best_cur = lk; ls_cur = lsk; }
            if (has_next && road == next_road && dist < d_next) { d_next = dist; best_next = lk; ls_next = lsk; }
        }
    }
    MD_FINE_STAMP(n == 0 && lane_id == 0, 3);
    if (lane_id != 0) return;  // everything below is wave-uniform; lane 0 commits
    int lane = -1;
    float ls = 0.0f;
    if (best_cur >= 0) { lane = best_cur; ls = ls_cur; }
    else if (!has_next) { lane = best_any; ls = ls_any; }
    else if (best_next >= 0) { lane = best_next; ls = ls_next; }
    else { lane = best_any; ls = ls_any; }
    if (onlane_out) {
        onlane_out[n] = on_lane ? MD_FL_ON_LANE : 0u;
    } else {
        uint32_t fl = s.flags[n] & ~(uint32_t)MD_FL_ON_LANE;
        if (on_lane) fl |= MD_FL_ON_LANE;
        s.flags[n] = fl;
    }
    const bool kept = lane < 0;  // found nothing: the previous lane stays, its longitudinal was not evaluated above
    if (kept) lane = nav.lane;
    s.nav[n].lane = lane;
    if (lane < 0) return;
    if (nav.ck0 == nav.ck1) return;
    if (kept) {
        float llat;
        md_lane_local(&lanes[lane], sh.cx, sh.cy, &ls, &llat);
    }
    if (!(ls < 5.0f)) return;
    const int start_node = roads[lanes[lane].road].start_node;
    const int k = nav.route_len;
    int idx = -1;
    for (int j = nav.ck1; j < k - 1; ++j) {
        if (rnodes[j] == start_node) { idx = j; break; }
    }
    if (idx < 0) return;
    const int nck1 = (idx + 1 == k - 1) ? idx : idx + 1;
    s.nav[n].ck0 = idx;
    s.nav[n].ck1 = nck1;
    s.nav[n].road0 = rroads[idx];
    s.nav[n].road1 = rroads[nck1];
}

// ------------------------------------------------------------------------------------------------
// Localisation of TWO vehicles by one wave (lanes 0-31: slot_a, lanes 32-63: slot_b; -1 = half idle): the form the
// fused step uses when more vehicles drive than the workgroup has waves.  Same arithmetic, same candidate order and
// tie-breaks as localize_vehicle; the per-vehicle data are per-lane values here (uniform within a half), cross-lane
// reads go through __shfl, votes through the matching half of a 64-bit ballot.  Results go to onlane_out.
// ------------------------------------------------------------------------------------------------
// GW = lanes per vehicle: 32 (two vehicles per wave) or 16 (four; the one-wave-per-env kernel's form when three or more
// vehicles drive).  slot_c / slot_d are only read with GW == 16.
template <int GW>
__device__ __forceinline__ void localize_group(const MdWorld& w, const MdLane* lanes, const MdRoad* roads, const MdState& s, int e,
                              int slot_a, int slot_b, int slot_c, int slot_d, int lane_id, uint32_t* onlane_out) {
    constexpr unsigned kGroupMask = (GW == 32) ? 0xFFFFFFFFu : 0xFFFFu;
    const int h = lane_id / GW, hl = lane_id & (GW - 1), hbase = h * GW;
    const int n = (h == 0) ? slot_a : ((h == 1) ? slot_b : ((h == 2) ? slot_c : slot_d));
    const bool act = n >= 0;  // (slots handed in always drive)
    const int nn = act ? n : 0;
    struct { float cx, cy, c, s; } sh;   // only what the search needs, per lane
    sh.cx = s.shape[nn].cx;
    sh.cy = s.shape[nn].cy;
    sh.c = s.shape[nn].c;
    sh.s = s.shape[nn].s;
    struct { int lane, ck0, ck1, route_len; } nav;
    nav.lane = s.nav[nn].lane;
    nav.ck0 = s.nav[nn].ck0;
    nav.ck1 = s.nav[nn].ck1;
    nav.route_len = s.nav[nn].route_len;
    const int m = w.env_map[e];
    const int32_t* rroads = s.route_roads + (size_t)nn * MD_ROUTE_LEN;   // global: read only when the cursors advance
    const int32_t* rnodes = s.route_nodes + (size_t)nn * MD_ROUTE_LEN;
    const int cur_road = s.nav[nn].road0;
    const bool has_next = nav.ck1 != nav.ck0;
    const int next_road = has_next ? s.nav[nn].road1 : -1;
    const MdGrid g = w.grid[m];
    const int gx = (int)md_floor((sh.cx - g.x0) * g.inv_cell);
    const int gy = (int)md_floor((sh.cy - g.y0) * g.inv_cell);
    int it0 = 0, it1 = 0;
    if (act && gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny) {
        const int cell = g.cell_base + gy * g.nx + gx;
        it0 = w.cell_start[cell];
        it1 = w.cell_start[cell + 1];
    }
    int on_lane = 0;
    int best_any = -1, best_cur = -1, best_next = -1;
    float d_any = 3.0e38f, d_cur = 3.0e38f, d_next = 3.0e38f;
    float ls_any = 0.0f, ls_cur = 0.0f, ls_next = 0.0f;
    const auto part_of = [&](unsigned long long b) { return (unsigned)(b >> hbase) & kGroupMask; };
    for (int itb = it0; __ballot(itb < it1) != 0ull; itb += GW) {
        const int it = itb + hl;
        int l = -1;
        bool pass = false;
        if (it < it1) {
            l = w.cell_items[it];
            if (l >= 0) {
                const MdLane* L = &lanes[l];
                pass = !(sh.cx < L->x0 || sh.cx > L->x1 || sh.cy < L->y0 || sh.cy > L->y1);
            }
        }
        unsigned mask = part_of(__ballot(pass));
        unsigned inside = 0u;  // bit k: candidate held by lane k of MY group contains the point
        while (__ballot(mask != 0u) != 0ull) {
            const bool mine = mask != 0u;
            const int k = mine ? (__ffs((int)mask) - 1) : 0;
            if (mine) mask &= mask - 1;
            const int lk = __shfl(l, hbase + k, 64);
            bool outside = false;
            int hn = 0;
            if (mine) {
                const MdLane* L = &lanes[lk];
                hn = L->hull_n;
                const float* xy = md_lane_hull(L, w.hull_xy);
                for (int i = hl; i < hn; i += GW) {
                    const int j = (i + 1 == hn) ? 0 : i + 1;
                    const float ex = xy[2 * j] - xy[2 * i], ey = xy[2 * j + 1] - xy[2 * i + 1];
                    const float cr = ex * (sh.cy - xy[2 * i + 1]) - ey * (sh.cx - xy[2 * i]);
                    if (cr < 0.0f) outside = true;
                }
            }
            const unsigned out_h = part_of(__ballot(outside));
            if (mine && out_h == 0u && hn >= 3) inside |= 1u << k;
        }
        if (inside != 0u) on_lane = 1;
        float my_dist = 3.0e38f, my_ls = 0.0f;
        int my_road = -1;
        if ((inside >> hl) & 1u) {
            const MdLane* L = &lanes[l];
            float llat;
            md_lane_local(L, sh.cx, sh.cy, &my_ls, &llat);
            if (md_lane_heading_dot(L, sh.cx, sh.cy, sh.c, sh.s) > 0.0f) {
                my_dist = md_lane_distance(L, my_ls, llat);
                my_road = L->road;
            }
        }
        while (__ballot(inside != 0u) != 0ull) {
            const bool mine = inside != 0u;
            const int k = mine ? (__ffs((int)inside) - 1) : 0;
            if (mine) inside &= inside - 1;
            const float dist = __shfl(my_dist, hbase + k, 64);
            const int road = __shfl(my_road, hbase + k, 64);
            const int lk = __shfl(l, hbase + k, 64);
            const float lsk = __shfl(my_ls, hbase + k, 64);
            if (!mine || road < 0) continue;
            if (dist < d_any) { d_any = dist; best_any = lk; ls_any = lsk; }
            if (road == cur_road && dist < d_cur) { d_cur = dist; best_cur = lk; ls_cur = lsk; }
            if (has_next && road == next_road && dist < d_next) { d_next = dist; best_next = lk; ls_next = lsk; }
        }
    }
    if (hl != 0 || !act) return;  // lane 0 of each group commits its vehicle
    int lane = -1;
    float ls = 0.0f;
    if (best_cur >= 0) { lane = best_cur; ls = ls_cur; }
    else if (!has_next) { lane = best_any; ls = ls_any; }
    else if (best_next >= 0) { lane = best_next; ls = ls_next; }
    else { lane = best_any; ls = ls_any; }
    onlane_out[n] = on_lane ? MD_FL_ON_LANE : 0u;
    const bool kept = lane < 0;  // found nothing: the previous lane stays, its longitudinal was not evaluated above
    if (kept) lane = nav.lane;
    s.nav[n].lane = lane;
    if (lane < 0) return;
    if (nav.ck0 == nav.ck1) return;
    if (kept) {
        float llat;
        md_lane_local(&lanes[lane], sh.cx, sh.cy, &ls, &llat);
    }
    if (!(ls < 5.0f)) return;
    const int start_node = roads[lanes[lane].road].start_node;
    const int kk = nav.route_len;
    int idx = -1;
    for (int j = nav.ck1; j < kk - 1; ++j) {
        if (rnodes[j] == start_node) { idx = j; break; }
    }
    if (idx < 0) return;
    const int nck1 = (idx + 1 == kk - 1) ? idx : idx + 1;
    s.nav[n].ck0 = idx;
    s.nav[n].ck1 = nck1;
    s.nav[n].road0 = rroads[idx];
    s.nav[n].road1 = rroads[nck1];
}

__device__ __forceinline__ void localize_pair(const MdWorld& w, const MdLane* lanes, const MdRoad* roads, const MdState& s, int e,
                              int slot_a, int slot_b, int lane_id, uint32_t* onlane_out) {
    localize_group<32>(w, lanes, roads, s, e, slot_a, slot_b, -1, -1, lane_id, onlane_out);
}

// ------------------------------------------------------------------------------------------------
// Contacts, one wave per vehicle.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void contacts_vehicle(const MdWorld& w, const MdState& s, const MdConfig& c, int e, int slot, int lane_id,
                                 uint32_t* cfl_out = nullptr) {
    const int base = 0;  // env-local view
    const int n = slot;
    const MdShape me = s.shape[n];
    if (!md_drives(me.flags)) return;
    if (!(me.flags & MD_F_AGENT)) {  // traffic: nothing reads its crash / line flags (see the oracle's contacts_mover)
        if (lane_id == 0) {
            if (cfl_out) cfl_out[n] = 0u;
            else s.flags[n] &= MD_FL_ON_LANE;
        }
        return;
    }
    uint32_t fl = 0;
    for (int j0 = 0; j0 < c.cap; j0 += 64) {
        const int j = j0 + lane_id;
        if (j >= c.cap || j == slot) continue;
        const MdShape o = s.shape[base + j];
        if (!md_present(o.flags)) continue;
        const int k = md_kind_of(o.flags);
        int hit;
        if (md_is_circle_kind(k)) hit = md_obb_circle(me.cx, me.cy, me.c, me.s, me.hl, me.hw, o.cx, o.cy, o.hl);
        else hit = md_obb_obb(me.cx, me.cy, me.c, me.s, me.hl, me.hw, o.cx, o.cy, o.c, o.s, o.hl, o.hw);
        if (!hit) continue;
        if (k == MD_KIND_VEHICLE) fl |= MD_FL_CRASH_VEHICLE;
        else if (k == MD_KIND_CONE || k == MD_KIND_WARNING || k == MD_KIND_BARRIER) fl |= MD_FL_CRASH_OBJECT;
        else if (k == MD_KIND_PEDESTRIAN || k == MD_KIND_CYCLIST) fl |= MD_FL_CRASH_HUMAN;
        else if (k == MD_KIND_BUILDING) fl |= MD_FL_CRASH_BUILDING;
    }
    // static quads through the grid: cells under the chassis AABB (+ margin)
    const int m = w.env_map[e];
    const MdGrid g = w.grid[m];
    const float ext_x = md_fabs(me.c) * me.hl + md_fabs(me.s) * me.hw + 0.05f;
    const float ext_y = md_fabs(me.s) * me.hl + md_fabs(me.c) * me.hw + 0.05f;
    int gx0 = (int)md_floor((me.cx - ext_x - g.x0) * g.inv_cell);
    int gx1 = (int)md_floor((me.cx + ext_x - g.x0) * g.inv_cell);
    int gy0 = (int)md_floor((me.cy - ext_y - g.y0) * g.inv_cell);
    int gy1 = (int)md_floor((me.cy + ext_y - g.y0) * g.inv_cell);
    if (!(gx1 < 0 || gy1 < 0 || gx0 >= g.nx || gy0 >= g.ny)) {
        gx0 = grid_clampi(gx0, 0, g.nx - 1);
        gx1 = grid_clampi(gx1, 0, g.nx - 1);
        gy0 = grid_clampi(gy0, 0, g.ny - 1);
        gy1 = grid_clampi(gy1, 0, g.ny - 1);
        const float* quads = w.quads + 8 * (size_t)w.quad_off[m];
        const int32_t* qkind = w.quad_kind + w.quad_off[m];
        for (int gy = gy0; gy <= gy1; ++gy)
            for (int gx = gx0; gx <= gx1; ++gx) {
                const int cell = g.cell_base + gy * g.nx + gx;
                const int it0 = w.cell_start[cell], it1 = w.cell_start[cell + 1];
                for (int it = it0 + lane_id; it < it1; it += 64) {
                    const int item = w.cell_items[it];
                    if (item >= 0) continue;  // lane item
                    const int q = ~item;
                    if (!md_obb_quad(me.cx, me.cy, me.c, me.s, me.hl, me.hw, quads + 8 * (size_t)q)) continue;
                    switch (qkind[q]) {
                        case MD_Q_LINE_WHITE_CONT: fl |= MD_FL_ON_WHITE_CONT; break;
                        case MD_Q_LINE_YELLOW_CONT: fl |= MD_FL_ON_YELLOW_CONT; break;
                        case MD_Q_LINE_BROKEN: fl |= MD_FL_ON_BROKEN; break;
                        case MD_Q_SIDEWALK: fl |= MD_FL_CRASH_SIDEWALK; break;
                        case MD_Q_CROSSWALK: fl |= MD_FL_ON_CROSSWALK; break;
                        default: break;
                    }
                }
            }
    }
    // wave OR-reduce of the flag word: one ballot per flag bit that can be set here
    uint32_t all = 0;
#pragma unroll
    for (int b = 0; b < 9; ++b) {
        const uint32_t bit = 1u << b;
        if (__ballot((fl & bit) != 0) != 0ull) all |= bit;
    }
    if (lane_id == 0) {
        if (cfl_out) cfl_out[n] = all;
        else s.flags[n] = (s.flags[n] & MD_FL_ON_LANE) | all;
    }
}

// ------------------------------------------------------------------------------------------------
// IDM for one traffic vehicle, executed by one wave: lanes = candidate objects; the lead / rear
// vehicle gap scan is a wavefront arg-min reduction (key = gap, tie -> lowest slot), which is the
// order-independent form of FrontBackObjects.get_find_front_back_objs (policy/idm_policy.py:82-132).
// Stages A (route bookkeeping) and C (lane-change policy, PID steering, IDM acceleration) are
// scalar and run on lane 0.
// ------------------------------------------------------------------------------------------------
// arg-min of (key, slot) over the wave; lanes holding kInf do not take part.  Ascending lane order
// == ascending slot order, so the strict `<` keeps the lowest slot among equal keys.
__device__ __forceinline__ void wave_argmin(float& key, int& slot) {
    unsigned long long m = __ballot(key < 3.0e38f);
    float bk = 3.0e38f;
    int bs = 0x7fffffff;
    while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        const float k = __shfl(key, l, 64);
        const int sl = __shfl(slot, l, 64);
        if (k < bk) {
            bk = k;
            bs = sl;
        }
    }
    key = bk;
    slot = bs;
}

// The object scan of one vehicle by the whole wave.  `plan` is wave-uniform (may come back with fail = 1: a traffic
// participant among the candidates), `fb` is the wave-uniform result.
// wave_list: >= 24 ints of LDS private to this wave (the compacted candidate list)
__device__ __forceinline__ void idm_scan_wave(const MdLane* lanes, const MdState& s, const MdConfig& c, int slot, int lane_id,
                              int* wave_list, MdIdmPlan& plan, FrontBack& fb) {
    constexpr float kInf = 3.0e38f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        fb.front[i] = fb.back[i] = -1;
        fb.exist[i] = plan.ids[i] >= 0 && !plan.fail;
        fb.front_d[i] = fb.back_d[i] = IDM_MAX_LONG_DIST;
    }
    if (!plan.fail) {
        const float px = s.shape[slot].cx, py = s.shape[slot].cy;
        // (1) ego longitudinal on the three scanned lanes: lanes 0..2 evaluate one each, then broadcast
        float my_cur = 0.0f;
        {
            const int my_id = (lane_id == 0) ? plan.ids[0] : ((lane_id == 1) ? plan.ids[1] : ((lane_id == 2) ? plan.ids[2] : -1));
            if (my_id >= 0) {
                float tmp;
                md_lane_local(&lanes[my_id], px, py, &my_cur, &tmp);
            }
        }
        float cur_long[3], left_long[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            cur_long[i] = bcast_f(my_cur, i);
            left_long[i] = (plan.ids[i] >= 0) ? lanes[plan.ids[i]].length - cur_long[i] : 0.0f;
        }
        // candidate objects (get_surrounding_objects: within 50 m, present, not the vehicle itself), compacted in
        // ascending slot order into wave_list: usually a handful, whatever the slot capacity is
        int n_cand = 0;
        bool sees_participant = false;  // a pedestrian / cyclist among them: the reference's object loop raises (md_idm_sees_participant)
        for (int j0 = 0; j0 < c.cap; j0 += 64) {
            const int j = j0 + lane_id;
            const bool is_c = j < c.cap && j != slot && md_idm_is_candidate(&s.shape[j < c.cap ? j : 0], px, py);
            const unsigned long long mk = __ballot(is_c);
            if (__ballot(is_c && md_is_participant_kind(md_kind_of(s.shape[j < c.cap ? j : 0].flags))) != 0ull) sees_participant = true;
            const int rank = n_cand + __popcll(mk & ((1ull << lane_id) - 1ull));
            if (is_c && rank < 21) wave_list[rank] = j;
            n_cand += __popcll(mk);
        }
        if (sees_participant) {  // wave-uniform: bare-except fallback, nothing is scanned
            plan.fail = 1;
            n_cand = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int npairs = n_cand * 3;
        if (npairs <= 63) {
            // (2) every (scanned lane i, candidate j) pair on its own lane of the wave: ONE Frenet evaluation
            //     per pair (on lane i for a same-lane object, on the object's lane for a connected one)
            float val = 0.0f;
            int meta = 0;  // bit0 same-lane, bit1 lane i precedes obj lane, bit2 obj lane precedes lane i, bits 4-5 i, bits 8.. j
            if (n_cand > 0) {
                const int p = lane_id;
                const int i = p / n_cand;
                const int j = (p < npairs) ? wave_list[p - i * n_cand] : 0;
                const int id_i = (i == 0) ? plan.ids[0] : ((i == 1) ? plan.ids[1] : plan.ids[2]);
                const float cur_i = (i == 0) ? cur_long[0] : ((i == 1) ? cur_long[1] : cur_long[2]);
                if (p < npairs && id_i >= 0) {
                    const int ol = md_obj_lane_of(&s, j);
                    const MdLane* L = &lanes[id_i];
                    int mt = (i << 4) | (j << 8);
                    // one Frenet evaluation per pair: on lane i for a same-lane object, else on the object's lane
                    const MdLane* EL = L;
                    if (ol == id_i) mt |= 1;
                    else if (ol >= 0) {
                        EL = &lanes[ol];
                        mt |= (md_lane_is_previous_of(L, EL) ? 2 : 0) | (md_lane_is_previous_of(EL, L) ? 4 : 0);
                    }
                    if (mt & 7) {
                        float os, ot;
                        md_lane_local(EL, s.shape[j].cx, s.shape[j].cy, &os, &ot);
                        val = (mt & 1) ? os - cur_i : os;
                    }
                    meta = mt;
                }
            }
            // (3) per scanned lane: arg-min reductions (same-lane objects first, connected lanes only if none)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (plan.ids[i] < 0) continue;  // wave-uniform
                int found_front = 0, found_back = 0;
                const bool mine = ((meta >> 4) & 3) == i;
                {
                    const bool same = mine && (meta & 1);
                    float kf = (same && val > 0.0f && val < IDM_MAX_LONG_DIST) ? val : kInf;
                    float kb = (same && val < 0.0f && md_fabs(val) < IDM_MAX_LONG_DIST) ? md_fabs(val) : kInf;
                    int jf = meta >> 8, jb = meta >> 8;
                    wave_argmin(kf, jf);
                    wave_argmin(kb, jb);
                    if (kf < fb.front_d[i]) { fb.front_d[i] = kf; fb.front[i] = jf; found_front = 1; }
                    if (kb < fb.back_d[i]) { fb.back_d[i] = kb; fb.back[i] = jb; found_back = 1; }
                }
                if (found_front && found_back) continue;
                float kf = kInf, kb = kInf;
                if (mine && !(meta & 1) && (meta & 6)) {
                    // md_fb_neighbour's choice: front if (need_front && L precedes OL), else back if (need_back && OL precedes L)
                    if (!found_front && (meta & 2)) {
                        const float lg = val + left_long[i];
                        if (lg > 0.0f && lg < IDM_MAX_LONG_DIST) kf = lg;
                    } else if (!found_back && (meta & 4)) {
                        const float lg = lanes[md_obj_lane_of(&s, meta >> 8)].length - val + cur_long[i];
                        if (lg < IDM_MAX_LONG_DIST) kb = lg;
                    }
                }
                int jf = meta >> 8, jb = meta >> 8;
                wave_argmin(kf, jf);
                wave_argmin(kb, jb);
                if (kf < fb.front_d[i]) { fb.front_d[i] = kf; fb.front[i] = jf; }
                if (kb < fb.back_d[i]) { fb.back_d[i] = kb; fb.back[i] = jb; }
            }
        } else {
            // general capacity: per scanned lane, lanes = objects (two passes, chunks of 64 objects)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (plan.ids[i] < 0) continue;  // wave-uniform
                const MdLane* L = &lanes[plan.ids[i]];
                int found_front = 0, found_back = 0;
                for (int j0 = 0; j0 < c.cap; j0 += 64) {
                    const int j = j0 + lane_id;
                    float kf = kInf, kb = kInf;
                    if (j < c.cap && j != slot && md_idm_is_candidate(&s.shape[j], px, py) &&
                        md_obj_lane_of(&s, j) == plan.ids[i]) {
                        const float lg = md_fb_same_lane_gap(L, cur_long[i], &s.shape[j]);
                        if (lg > 0.0f && lg < IDM_MAX_LONG_DIST) kf = lg;
                        if (lg < 0.0f && md_fabs(lg) < IDM_MAX_LONG_DIST) kb = md_fabs(lg);
                    }
                    int jf = j, jb = j;
                    wave_argmin(kf, jf);
                    wave_argmin(kb, jb);
                    if (kf < fb.front_d[i]) { fb.front_d[i] = kf; fb.front[i] = jf; found_front = 1; }
                    if (kb < fb.back_d[i]) { fb.back_d[i] = kb; fb.back[i] = jb; found_back = 1; }
                }
                if (found_front && found_back) continue;
                for (int j0 = 0; j0 < c.cap; j0 += 64) {
                    const int j = j0 + lane_id;
                    float kf = kInf, kb = kInf;
                    if (j < c.cap && j != slot && md_idm_is_candidate(&s.shape[j], px, py)) {
                        const int ol = md_obj_lane_of(&s, j);
                        if (ol >= 0 && ol != plan.ids[i]) {
                            float lg;
                            const int cls = md_fb_neighbour(L, &lanes[ol], cur_long[i], left_long[i], &s.shape[j],
                                                            !found_front, !found_back, &lg);
                            if (cls == 1 && lg > 0.0f && lg < IDM_MAX_LONG_DIST) kf = lg;
                            if (cls == 2 && lg < IDM_MAX_LONG_DIST) kb = lg;
                        }
                    }
                    int jf = j, jb = j;
                    wave_argmin(kf, jf);
                    wave_argmin(kb, jb);
                    if (kf < fb.front_d[i]) { fb.front_d[i] = kf; fb.front[i] = jf; }
                    if (kb < fb.back_d[i]) { fb.back_d[i] = kb; fb.back[i] = jb; }
                }
            }
        }
    }
}

// One vehicle at a time: plan on lane 0, scan by the wave, decide on lane 0 (the workgroup-per-env kernels' form).
__device__ __forceinline__ void idm_vehicle_wave(const MdWorld& w, const MdLane* lanes, const MdRoad* roads, const MdState& s,
                                 const MdConfig& c, int m, int slot, int lane_id, int* wave_list) {
    MdIdmPlan plan;
    plan.success = plan.use_ref = plan.fail = 0;
    plan.ids[0] = plan.ids[1] = plan.ids[2] = -1;
    const bool st_ = (slot == c.agents_per_env) && lane_id == 0;
    (void)st_;
    MD_FINE_STAMP(st_, 4);
    if (lane_id == 0) md_idm_plan(&w, lanes, roads, &s, &c, m, slot, &plan);
    plan.success = bcast_i(plan.success, 0);
    plan.use_ref = bcast_i(plan.use_ref, 0);
    plan.fail = bcast_i(plan.fail, 0);
    plan.ids[0] = bcast_i(plan.ids[0], 0);
    plan.ids[1] = bcast_i(plan.ids[1], 0);
    plan.ids[2] = bcast_i(plan.ids[2], 0);
    MD_FINE_STAMP(st_, 5);
    FrontBack fb;
    idm_scan_wave(lanes, s, c, slot, lane_id, wave_list, plan, fb);
    MD_FINE_STAMP(st_, 6);
    if (lane_id == 0) md_idm_decide(lanes, roads, &s, slot, &plan, &fb);
    MD_FINE_STAMP(st_, 7);
}

// All vehicles of `mask` (bit j = slot j0 + j; <= 64 of them): stage A (route bookkeeping) and stage C (lane-change
// policy, PID steering, IDM acceleration) are per-vehicle scalar code, so every vehicle runs them on ITS OWN lane, all
// at once; only the object scans in between go vehicle by vehicle, each by the whole wave.  Same arithmetic per
// vehicle; the decisions do not depend on each other (a decision reads poses / speeds / lanes, and writes only its
// own action, PID and lane-change state).  The one-wave-per-env kernel's form.
__device__ __forceinline__ void idm_group_wave(const MdWorld& w, const MdLane* lanes, const MdRoad* roads, const MdState& s,
                               const MdConfig& c, int m, int j0, unsigned long long mask, int lane_id, int* wave_list) {
    const int my_slot = j0 + lane_id;
    const bool mine = (mask >> lane_id) & 1ull;
    MdIdmPlan plan;
    plan.success = plan.use_ref = plan.fail = 0;
    plan.ids[0] = plan.ids[1] = plan.ids[2] = -1;
    if (mine) md_idm_plan(&w, lanes, roads, &s, &c, m, my_slot, &plan);
    FrontBack fb;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        fb.front[i] = fb.back[i] = -1;
        fb.exist[i] = 0;
        fb.front_d[i] = fb.back_d[i] = IDM_MAX_LONG_DIST;
    }
    unsigned long long todo = mask;
    while (todo) {
        const int v = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        MdIdmPlan up;   // vehicle v's plan, wave-uniform
        up.success = bcast_i(plan.success, v);
        up.use_ref = bcast_i(plan.use_ref, v);
        up.fail = bcast_i(plan.fail, v);
        up.ids[0] = bcast_i(plan.ids[0], v);
        up.ids[1] = bcast_i(plan.ids[1], v);
        up.ids[2] = bcast_i(plan.ids[2], v);
        FrontBack ufb;
        idm_scan_wave(lanes, s, c, j0 + v, lane_id, wave_list, up, ufb);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next scan reuses wave_list
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane_id == v) {
            plan.fail = up.fail;
            fb = ufb;
        }
    }
    if (mine) md_idm_decide(lanes, roads, &s, my_slot, &plan, &fb);
}

// ------------------------------------------------------------------------------------------------
// Observation / reward / done of one agent by one wave: the nine independent geometric evaluations
// (md_observe_task) run on lanes 0..8 at once, lane 0 gathers them with v_readlane and combines.
// ------------------------------------------------------------------------------------------------
// One agent per wave (`a` wave-uniform: its context lives in scalar registers) -- the single-agent envs' form.
__device__ __forceinline__ void observe_agent_wave1(const MdLane* lanes, const MdRoad* roads, const MdState& s, const MdConfig& c, int a,
                                    int just_reset, int lane_id, float* wave_scratch /* LDS, MD_OBS_TASKS*5 floats */) {
    MdObsCtx k;
    MD_FINE_STAMP(a == 0 && lane_id == 0, 8);
    md_observe_ctx(lanes, roads, &s, a, &k);
    MD_FINE_STAMP(a == 0 && lane_id == 0, 9);
    if (lane_id < MD_OBS_TASKS) {
        float mine[5];
        md_observe_task(lane_id, &k, &s, &c, a, mine);
#pragma unroll
        for (int i = 0; i < 5; ++i) wave_scratch[lane_id * 5 + i] = mine[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    MD_FINE_STAMP(a == 0 && lane_id == 0, 10);
    if (lane_id == 0) md_observe_combine(&k, &s, &c, a, just_reset, (const float (*)[5])wave_scratch);
    __builtin_amdgcn_wave_barrier();
    MD_FINE_STAMP(a == 0 && lane_id == 0, 11);
}

// Up to FOUR agents per wave: 16-lane group g serves agent a_base + g (tasks on its lanes 0..8, combine on its
// lane 0): a 40-agent env finishes in 3 rounds of 4 waves instead of 10 (multi-agent kernels only).
constexpr int kObsGroups = 4;
constexpr int kObsScratch = kObsGroups * 48;  // floats of LDS per wave: MD_OBS_TASKS * 5 (= 45) per group, padded

__device__ __forceinline__ void observe_agent_wave(const MdLane* lanes, const MdRoad* roads, const MdState& s, const MdConfig& c, int a_base,
                                   int just_reset, int lane_id, float* wave_scratch /* LDS, kObsScratch floats */) {
    const int g = lane_id >> 4, sub = lane_id & 15;
    const int a = a_base + g;
    const bool live = a < c.agents_per_env;
    float* scratch = wave_scratch + g * 48;
    MdObsCtx k;
    if (live) md_observe_ctx(lanes, roads, &s, a, &k);
    if (live && sub < MD_OBS_TASKS) {
        float mine[5];
        md_observe_task(sub, &k, &s, &c, a, mine);
#pragma unroll
        for (int i = 0; i < 5; ++i) scratch[sub * 5 + i] = mine[i];
    }
    // same wave: the LDS unit executes a wave's ds_write / ds_read in order; the fence keeps the
    // compiler from moving the combining lanes' reads above the other lanes' writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (live && sub == 0) md_observe_combine(&k, &s, &c, a, just_reset, (const float (*)[5])scratch);
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// Traffic trigger (wave 0) -- PGTrafficManager.before_step, manager/traffic_manager.py:80-88
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void trigger_env(const MdLane* lanes, const MdState& s, const MdConfig& c, int lane_id) {
    const int base = 0;  // env-local view
    int my_min = 0x7fffffff;
    for (int j = lane_id; j < c.cap; j += 64) {
        const int f = s.shape[base + j].flags;
        if ((f & MD_F_PENDING) && (f & MD_F_ALIVE)) {
            const int o = s.nav[base + j].trigger_order;
            my_min = o < my_min ? o : my_min;
        }
    }
    const int min_order = wave_min_i(my_min, my_min != 0x7fffffff, 0x7fffffff);  // wavefront min-reduce
    if (min_order == 0x7fffffff) return;
    // trigger road of that block: lowest slot carrying min_order
    int my_slot = 0x7fffffff;
    for (int j = lane_id; j < c.cap; j += 64) {
        const int f = s.shape[base + j].flags;
        if ((f & MD_F_PENDING) && (f & MD_F_ALIVE) && s.nav[base + j].trigger_order == min_order)
            my_slot = j < my_slot ? j : my_slot;
    }
    const int first_slot = wave_min_i(my_slot, my_slot != 0x7fffffff, 0x7fffffff);
    const int trig_road = s.nav[base + first_slot].trigger_road;
    bool fire = false;
    for (int a = lane_id; a < c.agents_per_env; a += 64) {
        if (!md_drives(s.shape[base + a].flags)) continue;
        const int al = s.nav[base + a].lane;
        if (al >= 0 && lanes[al].road == trig_road) fire = true;
    }
    if (__ballot(fire) == 0ull) return;
    for (int j = lane_id; j < c.cap; j += 64) {
        const int f = s.shape[base + j].flags;
        if ((f & MD_F_PENDING) && (f & MD_F_ALIVE) && s.nav[base + j].trigger_order == min_order)
            s.shape[base + j].flags = f & ~MD_F_PENDING;
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-agent lifecycle of one env by the whole workgroup: the block-parallel form of md_lifecycle_env
// (include/md_entity.h, which the oracle runs serially).  Per-agent state machine: one thread per agent; the
// search for a safe spawn place -- (place, vehicle) overlap tests, 8 x 40 of them -- spread over all threads with
// an LDS OR per place; the respawn itself (RNG draws, slot rewrite) on thread 0.  md_lifecycle_env re-scans
// after a respawn; that second scan can never find a place (every safe place of the first scan is marked used,
// the unsafe ones are still occupied), so one scan is exact.  scratch: >= 4 ints of LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lifecycle_block(const MdWorld& w, const MdState& s, const MdConfig& c, int m, int tid, int nthreads,
                                int* scratch) {
    const int A = c.agents_per_env;
    int* cnt_active = scratch;      // agents that keep driving
    int* cnt_dying = scratch + 1;   // bodies waiting out delay_done
    int* hit_mask = scratch + 2;    // bit p: spawn place p is occupied
    int* reserved = scratch + 3;    // parking lot env: bit d: an active agent holds parking space d (md_lifecycle_env)
    int* new_slot = scratch + 4;    // slot refilled in this step (-1: none) and its row of the spawn-route table: the
    int* new_ri = scratch + 5;      //   96 words of its route are copied by the whole workgroup, not by the one thread
    const bool parking = c.ma_kind == MD_MA_PARKING_LOT;
    if (tid == 0) {
        s.env_steps[0] += 1;
        *cnt_active = 0;
        *cnt_dying = 0;
        *hit_mask = 0;
        *reserved = 0;
        *new_slot = -1;
    }
    __syncthreads();
    for (int a = tid; a < A; a += nthreads) {
        MdShape* sh = &s.shape[a];
        MdNav* nav = &s.nav[a];
        if (!(sh->flags & MD_F_ALIVE)) continue;
        bool count_down = true;
        if (!(sh->flags & MD_F_STATIC)) {
            const uint32_t fl = s.flags[a];
            if (nav->done || (fl & MD_FL_TRUNCATED)) {
                if ((fl & MD_FL_ARRIVE_DEST) || c.delay_done <= 0) {
                    sh->flags &= ~MD_F_ALIVE;
                    count_down = false;
                } else {
                    sh->flags |= MD_F_STATIC;
                    nav->timer = c.delay_done;
                    s.dyn[a].speed = 0.0f;
                }
            } else {
                atomicAdd(cnt_active, 1);
                if (parking && nav->toll_entry > 0) atomicOr(reserved, 1 << (nav->toll_entry - 1));
                count_down = false;
            }
        }
        if (count_down) {
            nav->timer -= 1;
            if (nav->timer <= 0) sh->flags &= ~MD_F_ALIVE;
            else atomicAdd(cnt_dying, 1);
        }
    }
    __syncthreads();
    const bool horizon_open = !(c.horizon > 0 && s.env_steps[0] >= c.horizon);
    const bool may_respawn = c.allow_respawn && horizon_open && w.spawn_off != nullptr && (*cnt_active + *cnt_dying < A);
    if (may_respawn) {  // block-uniform
        const int p0 = w.spawn_off[m];
        const int np_ = min(w.spawn_off[m + 1] - p0, 32);
        for (int idx = tid; idx < np_ * c.cap; idx += nthreads) {
            const int p = idx / c.cap, j = idx - p * c.cap;
            const MdShape o = s.shape[j];
            if (!md_present(o.flags) || md_kind_of(o.flags) != MD_KIND_VEHICLE) continue;
            const float* pl = w.spawn_place + 8 * (size_t)(p0 + p);
            if (md_obb_obb(pl[0], pl[1], pl[2], pl[3], MD_RESPAWN_HALF_LEN, MD_RESPAWN_HALF_WID, o.cx, o.cy, o.c, o.s, o.hl, o.hw))
                atomicOr(hit_mask, 1 << p);
        }
        __syncthreads();
        if (tid == 0) {
            const uint32_t all = np_ >= 32 ? 0xFFFFFFFFu : ((1u << np_) - 1u);
            uint32_t safe = ~(uint32_t)(*hit_mask) & all;
            const int n_in = parking ? np_ - c.n_parking : 0;
            uint32_t avail = parking ? (((1u << c.n_parking) - 1u) & ~(uint32_t)(*reserved)) : 0u;
            if (parking && avail == 0u) safe &= ~((1u << n_in) - 1u);   // no free space: the entrances stay shut
            const int n_safe = __popc(safe);
            int slot = -1;
            for (int a = 0; a < A; ++a)
                if (!(s.shape[a].flags & MD_F_ALIVE)) { slot = a; break; }
            if (n_safe > 0) {
                int k = (int)(md_rng_next(s.rng) % (uint32_t)n_safe);   // ascending place order, like the serial `safe[]`
                while (k-- > 0) safe &= safe - 1;
                const int p = __ffs((int)safe) - 1;
                if (slot >= 0) {
                    const float* pl = w.spawn_place + 8 * (size_t)(p0 + p);
                    int dest = 0, space = 0;
                    if (!parking) dest = (int)(md_rng_next(s.rng) % (uint32_t)w.n_dest);
                    else if (p < n_in) {
                        dest = md_kth_set_bit(avail, (int)(md_rng_next(s.rng) % (uint32_t)__popc(avail)));
                        space = dest + 1;
                    } else dest = c.n_parking + (int)(md_rng_next(s.rng) % (uint32_t)(w.n_dest - c.n_parking));
                    const size_t ri = ((size_t)(p0 + p) * w.n_dest + dest);
                    const int32_t* rt = w.spawn_route + ri * 2 * MD_ROUTE_LEN;
                    MdShape* sh = &s.shape[slot];
                    sh->cx = pl[0];
                    sh->cy = pl[1];
                    sh->c = pl[2];
                    sh->s = pl[3];
                    sh->flags = MD_KIND_VEHICLE | MD_F_ALIVE | MD_F_AGENT | MD_F_SPAWNED;
                    sh->aux = -1;
                    MdDyn* d = &s.dyn[slot];
                    d->heading = pl[4];
                    d->speed = 0.0f;
                    d->steering = 0.0f;
                    d->throttle = 0.0f;
                    d->last_x = pl[0];
                    d->last_y = pl[1];
                    d->last_c = pl[2];
                    d->last_s = pl[3];
                    MdNav* nav = &s.nav[slot];
                    nav->lane = w.spawn_lane[p0 + p];
                    nav->route_len = w.spawn_route_meta[2 * ri];
                    nav->ck0 = 0;
                    nav->ck1 = (nav->route_len <= 2) ? 0 : 1;
                    nav->target_lane = -1;
                    nav->timer = 0;
                    nav->steps = 0;
                    nav->done = 0;
                    nav->toll_state = nav->toll_entry = nav->toll_exit = 0;
                    nav->toll_entry = space;
                    md_agent_idm_init(&s, &c, slot);
                    if (c.random_agent_model && w.n_vclass > 0) md_draw_vehicle_class(&w, &s, slot);
                    s.final_lane[slot] = w.spawn_route_meta[2 * ri + 1];
                    *new_slot = slot;
                    *new_ri = (int)ri;
                    nav->road0 = rt[MD_ROUTE_LEN + nav->ck0];
                    nav->road1 = rt[MD_ROUTE_LEN + nav->ck1];
                    s.pid[slot].energy = 0.0f;
                    s.flags[slot] = 0;
                    s.action[2 * slot] = 0.0f;
                    s.action[2 * slot + 1] = 0.0f;
                    s.agent_id[slot] = s.next_agent_id[0];
                    s.next_agent_id[0] += 1;
                    *cnt_active += 1;
                }
            }
        }
        __syncthreads();
        if (*new_slot >= 0) {   // block-uniform
            const int slot = *new_slot;
            const int32_t* rt = w.spawn_route + (size_t)(*new_ri) * 2 * MD_ROUTE_LEN;
            for (int q = tid; q < 2 * MD_ROUTE_LEN; q += nthreads) {
                if (q < MD_ROUTE_LEN) s.route_nodes[(size_t)slot * MD_ROUTE_LEN + q] = rt[q];
                else s.route_roads[(size_t)slot * MD_ROUTE_LEN + (q - MD_ROUTE_LEN)] = rt[q];
            }
        }
    }
    // episode over: nobody left and nobody can come back
    if (tid == 0 && c.auto_reset && *cnt_active == 0 && !(c.allow_respawn && horizon_open)) s.need_reset[0] = 1;
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// The per-env kernel.  PH selects the phases (a compile-time mask: the single-phase entry points
// instantiate it with one bit, md_step with all of them).
// ------------------------------------------------------------------------------------------------
// Cooperative 16-byte copy (both sides 16-byte aligned, nbytes a multiple of 16).
__device__ __forceinline__ void copy16(void* dst, const void* src, int nbytes, int tid, int nthreads) {
    uint4* d = reinterpret_cast<uint4*>(dst);
    const uint4* s = reinterpret_cast<const uint4*>(src);
    for (int i = tid; i < (nbytes >> 4); i += nthreads) d[i] = s[i];
}

// STAGE_MAP: copy the env's lane + road tables into LDS (small maps: <= kStageMaxLanes lanes); big maps
// (intersections, roundabouts: 100-200 lanes) are read through L1/L2 instead -- staging 30 KB per env per
// step would cost more HBM traffic and LDS occupancy than it saves in latency.
constexpr int kStageMaxLanes = 64;
constexpr int kWideMaxEnvs = 640;   // multi-agent batches up to this many envs step with eight waves per env (see launch<>)
constexpr uint32_t kRemovedMark = 0xFFFFFFFFu;  // l_cfl value of a traffic slot removed in this step

// RESPAWN: the variant for everything off the headline path -- traffic modes respawn / hybrid / replay, detected
// sets for the `num_others` block (compiled apart: its slot-rewriting code costs the common trigger-mode
// kernel 8 VGPRs and one wave of occupancy when it is merely branched around).
// Threads per env workgroup.  256 = 4 waves: measured best (tools/run_blocks.sh rebuilds with -DMD_ENV_BLOCK=128/64).
#ifndef MD_ENV_BLOCK
#define MD_ENV_BLOCK 256
#endif
#ifndef MD_ENV_WAVES_EU
#define MD_ENV_WAVES_EU 7
#endif
// MULTI: multi-agent envs (lifecycle phase, reference order of the IDM); single-agent envs plan the traffic ahead.
// Register budget: the single-agent fused step is compiled for 7 waves per SIMD (72 VGPRs / 96 SGPRs): measured
// 123 us against 133 us at the compiler's own choice (6 waves) and 131 us at 8 (64 VGPRs, more spills) --
// the step is latency-bound, so resident workgroups per CU count.  Other instantiations keep the default.
template <int PH, bool RESPAWN, bool MULTI>
constexpr int env_waves_per_eu() { return (PH == PH_ALL && !RESPAWN && !MULTI && MD_ENV_BLOCK >= 128) ? MD_ENV_WAVES_EU : 0; }

// BLK: threads of the workgroup (MD_ENV_BLOCK; multi-agent batches small enough to stay resident are also instantiated with 512:
// eight waves share an env's 20-40 agents -- lifecycle search, contacts, observe groups, lidar sectors -- instead of four).
template <int PH, bool STAGE_MAP, bool RESPAWN = false, bool MULTI = false, int BLK = MD_ENV_BLOCK>
__global__ __launch_bounds__(BLK)
__attribute__((amdgpu_waves_per_eu(env_waves_per_eu<PH, RESPAWN, MULTI>() ? env_waves_per_eu<PH, RESPAWN, MULTI>() : 1,
                                   env_waves_per_eu<PH, RESPAWN, MULTI>() ? env_waves_per_eu<PH, RESPAWN, MULTI>() : 8)))
void env_kernel(MdWorld w, MdState g, MdConfig c, float* lidar_out,
                                                  int lidar_stride, int lidar_offset) {
    constexpr int kBlock = BLK;
    constexpr int kWaves = kBlock / 64;
    if ((int)blockIdx.x >= c.n_envs) return;
#ifdef MD_STAMP
    const int e = g_env_order ? g_env_order[blockIdx.x] : (int)blockIdx.x;   // diagnostic: launch-order experiments
#else
    const int e = blockIdx.x;
#endif
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int cap = c.cap;

    // ---- LDS image of this env's dynamic state (dynamic LDS: cap * 172 B + 16 B) ----
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    MdShape* l_shape = reinterpret_cast<MdShape*>(smem);
    MdDyn* l_dyn = reinterpret_cast<MdDyn*>(l_shape + cap);
    MdNav* l_nav = reinterpret_cast<MdNav*>(l_dyn + cap);
    MdPid* l_pid = reinterpret_cast<MdPid*>(l_nav + cap);
    float* l_action = reinterpret_cast<float*>(l_pid + cap);
    uint32_t* l_flags = reinterpret_cast<uint32_t*>(l_action + 2 * cap);
    // static tables of this env's map: read many times by the serial per-vehicle logic, so (small maps) they sit in
    // LDS too (a dependent chain of HBM/L2 round trips otherwise).  The movers' routes stay in global memory: the two
    // road ids the step needs are cached in MdNav (road0 / road1), the arrays are touched only when a cursor advances.
    const int n_stage_lanes = STAGE_MAP ? w.max_lanes : 0, n_stage_roads = STAGE_MAP ? w.max_roads : 0;
    MdLane* l_lanes = reinterpret_cast<MdLane*>(l_flags + ((cap + 3) & ~3));
    MdRoad* l_roads = reinterpret_cast<MdRoad*>(l_lanes + n_stage_lanes);
    constexpr int kScratch = MULTI ? kObsScratch : 48;  // floats per wave (launch<> sizes the LDS image the same way)
    float* l_scratch = reinterpret_cast<float*>(l_roads + n_stage_roads) + wave * kScratch;  // per-wave observe results
    MdParam* l_param = reinterpret_cast<MdParam*>(reinterpret_cast<float*>(l_roads + n_stage_roads) + kWaves * kScratch);
    int32_t* l_final = reinterpret_cast<int32_t*>(l_param + cap);
    unsigned long long* l_det = reinterpret_cast<unsigned long long*>(l_final + ((cap + 1) & ~1));  // [A][2] detected sets
    uint32_t* l_onlane = reinterpret_cast<uint32_t*>(l_det + 2 * c.agents_per_env);  // fused step: localize / contacts results,
    uint32_t* l_cfl = l_onlane + cap;                                                  // merged into flags afterwards
    // MULTI: ticket counters of the locate / observe / lidar stages -- behind the lifecycle's 8 scratch words, which start at l_onlane
    // and reach past l_cfl + cap in an env of fewer than four slots
    int* l_tk = reinterpret_cast<int*>(l_onlane + (2 * cap > 8 ? 2 * cap : 8));
    // detected sets: only the RESPAWN ("everything else") and MULTI variants carry the tracking code; launch<> picks
    // one of them whenever MdState.detected is set, so the lean trigger-mode kernel pays nothing for it
    const bool track_det = (PH == PH_ALL) && (RESPAWN || MULTI) && g.detected != nullptr;

    const MdState gv = md_env_view(&g, &c, e);  // this env's slices of the global arrays
    constexpr bool kLidarOnly = (PH == PH_LIDAR);
    // The reset flag is loaded first but nothing below waits for it: the live state is staged
    // unconditionally (one round trip) and only a resetting env re-stages from the snapshot.
    const int reset_flag = (PH & PH_RESET) ? gv.need_reset[0] : 0;

    MD_STAMP_AT(0);
    const bool kFusedAct = (PH == PH_ALL) && gv.agent_action != nullptr;  // md_step only: agents' actions from the caller's buffer
    const MdLane* lanes;
    const MdRoad* roads;
    if (STAGE_MAP) {
        lanes = l_lanes;
        roads = l_roads;
    } else {
        lanes = w.lanes + w.lane_off[w.env_map[e]];
        roads = w.roads + w.road_off[w.env_map[e]];
    }
    if (kLidarOnly) {
        copy16(l_shape, gv.shape, cap * (int)sizeof(MdShape), tid, kBlock);
    } else if (kBlock >= 256) {
        // Fast path: ALL global loads of the stage-in are issued before the first LDS store, so the whole image
        // arrives in one memory round trip (a chain of copy loops waits for each loop's loads in turn: 7 trips,
        // ~9 k cycles per env).  16-B units per array at cap <= 128: shape/dyn/pid/param <= 256 (one per thread),
        // nav <= 512, routes <= 1536: the first 256 / 512 units go through registers, the rest (cap > 64 / > 42) in tail loops.
        const int n32 = cap * 2, n64 = cap * 4;
        const uint4* g_shape = reinterpret_cast<const uint4*>(gv.shape);
        const uint4* g_dyn = reinterpret_cast<const uint4*>(gv.dyn);
        const uint4* g_pid = reinterpret_cast<const uint4*>(gv.pid);
        const uint4* g_param = reinterpret_cast<const uint4*>(gv.param);
        const uint4* g_nav = reinterpret_cast<const uint4*>(gv.nav);
        uint4 r_shape, r_dyn, r_pid, r_param, r_nav0;
        float2 r_act;
        uint32_t r_fl;
        int r_fin;
        const bool p32 = tid < n32, pc = tid < cap;
        if (p32) {
            r_shape = ld_stream(&g_shape[tid]);
            r_dyn = ld_stream(&g_dyn[tid]);
            r_pid = ld_stream(&g_pid[tid]);
            r_param = ld_stream(&g_param[tid]);
        }
        if (tid < n64) r_nav0 = ld_stream(&g_nav[tid]);
        if (pc) {
            r_act = (kFusedAct && tid < c.agents_per_env) ? reinterpret_cast<const float2*>(gv.agent_action)[tid]
                                                          : reinterpret_cast<const float2*>(gv.action)[tid];
            r_fl = gv.flags[tid];
            r_fin = gv.final_lane ? gv.final_lane[tid] : 0;
        }
        if (STAGE_MAP) {
            const int m = w.env_map[e];
            const int lo = w.lane_off[m], ro = w.road_off[m];
            copy16(l_lanes, w.lanes + lo, (w.lane_off[m + 1] - lo) * (int)sizeof(MdLane), tid, kBlock);
            copy16(l_roads, w.roads + ro, (w.road_off[m + 1] - ro) * (int)sizeof(MdRoad), tid, kBlock);
        }
        if (p32) {
            reinterpret_cast<uint4*>(l_shape)[tid] = r_shape;
            reinterpret_cast<uint4*>(l_dyn)[tid] = r_dyn;
            reinterpret_cast<uint4*>(l_pid)[tid] = r_pid;
            reinterpret_cast<uint4*>(l_param)[tid] = r_param;
        }
        if (tid < n64) reinterpret_cast<uint4*>(l_nav)[tid] = r_nav0;
        for (int i = tid + kBlock; i < n64; i += kBlock) reinterpret_cast<uint4*>(l_nav)[i] = g_nav[i];
        if (pc) {
            reinterpret_cast<float2*>(l_action)[tid] = r_act;
            l_flags[tid] = r_fl;
            l_final[tid] = r_fin;
        }
        if (track_det)
            for (int j = tid; j < 2 * c.agents_per_env; j += kBlock) l_det[j] = 0ull;
    } else {
        copy16(l_shape, gv.shape, cap * (int)sizeof(MdShape), tid, kBlock);
        if (STAGE_MAP) {
            const int m = w.env_map[e];
            const int lo = w.lane_off[m], ro = w.road_off[m];
            copy16(l_lanes, w.lanes + lo, (w.lane_off[m + 1] - lo) * (int)sizeof(MdLane), tid, kBlock);
            copy16(l_roads, w.roads + ro, (w.road_off[m + 1] - ro) * (int)sizeof(MdRoad), tid, kBlock);
        }
        copy16(l_dyn, gv.dyn, cap * (int)sizeof(MdDyn), tid, kBlock);
        copy16(l_nav, gv.nav, cap * (int)sizeof(MdNav), tid, kBlock);
        copy16(l_pid, gv.pid, cap * (int)sizeof(MdPid), tid, kBlock);
        copy16(l_param, gv.param, cap * (int)sizeof(MdParam), tid, kBlock);
        for (int j = tid; j < cap; j += kBlock) {
            const float* src = (kFusedAct && j < c.agents_per_env) ? gv.agent_action : gv.action;
            l_action[2 * j] = src[2 * j];
            l_action[2 * j + 1] = src[2 * j + 1];
            l_flags[j] = gv.flags[j];
            l_final[j] = gv.final_lane ? gv.final_lane[j] : 0;
        }
        if (track_det)
            for (int j = tid; j < 2 * c.agents_per_env; j += kBlock) l_det[j] = 0ull;
    }
    const bool do_reset = reset_flag != 0;  // block-uniform
    const int just_reset = do_reset ? 1 : 0;
    if (do_reset && !kLidarOnly) {
        __syncthreads();
        copy16(l_shape, gv.shape0, cap * (int)sizeof(MdShape), tid, kBlock);
        copy16(l_dyn, gv.dyn0, cap * (int)sizeof(MdDyn), tid, kBlock);
        copy16(l_nav, gv.nav0, cap * (int)sizeof(MdNav), tid, kBlock);
        copy16(l_pid, gv.pid0, cap * (int)sizeof(MdPid), tid, kBlock);
        if (MULTI && c.random_agent_model && gv.param0)   // respawns drew new vehicle classes: back to the reset ones
            copy16(l_param, gv.param0, cap * (int)sizeof(MdParam), tid, kBlock);
        for (int j = tid; j < cap; j += kBlock) {
            l_action[2 * j] = 0.0f;
            l_action[2 * j + 1] = 0.0f;
            l_flags[j] = 0u;
        }
        if (MULTI || (RESPAWN && (c.traffic_mode == 1 || c.traffic_mode == 2))) {  // respawns rewrote the routes: restore them too
            for (int i = tid; i < cap * MD_ROUTE_LEN; i += kBlock) {
                gv.route_nodes[i] = gv.route_nodes0[i];
                gv.route_roads[i] = gv.route_roads0[i];
            }
            for (int j = tid; j < cap; j += kBlock) l_final[j] = gv.final_lane0[j];
        }
        if (MULTI) {
            for (int j = tid; j < cap; j += kBlock) gv.agent_id[j] = j;
            if (tid == 0) {
                gv.env_steps[0] = 0;
                int n0 = 0;  // agents present at reset: all slots, or one per spawn point with num_agents = -1
                for (int j = 0; j < c.agents_per_env; ++j) n0 += (gv.shape0[j].flags & MD_F_ALIVE) ? 1 : 0;
                gv.next_agent_id[0] = n0;
            }
        }
    }
    MdState s = gv;  // env-local view whose hot arrays live in LDS
    s.shape = l_shape;
    if (!kLidarOnly) {
        s.dyn = l_dyn;
        s.nav = l_nav;
        s.pid = l_pid;
        s.action = l_action;
        s.flags = l_flags;
        s.param = l_param;
        s.final_lane = l_final;
    }
    if ((MULTI || (MD_LEAN_LIDAR_TICKETS && PH == PH_ALL)) && tid == 0) l_tk[0] = l_tk[1] = l_tk[2] = 0;
    __syncthreads();
    MD_STAMP_AT(1);

    if ((PH & PH_LIFECYCLE) && MULTI && !just_reset) {
        lifecycle_block(w, s, c, w.env_map[e], tid, kBlock, reinterpret_cast<int*>(l_onlane));
    }

    // Single-agent envs plan the traffic one step AHEAD (see the observe stage below): the IDM decision of step
    // t+1 depends only on the state at the end of step t, so it is taken there, on the waves that would
    // otherwise idle behind the agent's observe chain.  Multi-agent envs keep the reference order (the
    // lifecycle at the start of a step may respawn agents the IDM would have to see).
    constexpr bool kFused = (PH == PH_ALL);
    // agent_policy = IDMPolicy (single-agent envs): the agents are planned like the traffic, in the reference's order
    // (decide, then move): the observation reports the action applied in THIS step.  Never in the lean fused variant:
    // the launcher sends such configs to the RESPAWN one.
    const bool agent_idm = (kFused && !RESPAWN && !MULTI) ? false : (c.agent_idm != 0);
    constexpr bool kPlanAhead = kFused && kWaves > 1 && !MULTI;
    const bool plan_ahead = kPlanAhead && !agent_idm;
    if ((PH & PH_IDM) && !just_reset && !plan_ahead) {
        if (wave == 0) trigger_env(lanes, s, c, lane);
        __syncthreads();
        MD_STAMP_AT(2);
        for (int j = (agent_idm ? 0 : c.agents_per_env) + wave; j < cap; j += kWaves) {
            const int f = s.shape[j].flags;  // wave-uniform
            if (md_drives(f) && (agent_idm || !(f & MD_F_AGENT))) idm_vehicle_wave(w, lanes, roads, s, c, w.env_map[e], j, lane, reinterpret_cast<int*>(l_scratch));
        }
        __syncthreads();
    }
    MD_STAMP_AT(3);
    if ((PH & PH_INTEGRATE) && !just_reset) {
        for (int j = tid; j < cap; j += kBlock) {
            if (RESPAWN) md_advance_mover(&s, &c, j);  // the non-trigger traffic modes' kernel (respawn / hybrid / replay)
            else md_integrate_mover(&s, &c, j);
        }
        if (!RESPAWN)  // user-spawned pedestrians / cyclists (a loop of its own: inside the one above it costs spills)
            for (int j = tid; j < cap; j += kBlock) md_walk_mover(&s, &c, j);
        __syncthreads();
    }
    MD_STAMP_AT(4);
    unsigned long long drv_lo = 0ull, drv_hi = 0ull;  // fused step: the slots that drive in this step (wave-uniform)
    if (kFused) {
        // Localisation and contacts of a vehicle are independent of each other (both only read the integrated
        // poses and the map), so they share ONE stage: work items = [localize of every driving vehicle, then
        // contacts of every driving vehicle], dealt round-robin to the waves.  With the usual 1-3 driving
        // vehicles per env everything fits one round instead of two phases behind two barriers.
        for (int j0 = 0; j0 < cap; j0 += 64) {
            const int j = j0 + lane;
            const unsigned long long mk = __ballot(j < cap && md_drives(s.shape[j < cap ? j : 0].flags));
            if (j0 == 0) drv_lo = mk;
            else drv_hi = mk;
        }
        // contacts only for agents (traffic's crash flags are never read): their slots are the first A
        const int A_ = c.agents_per_env;
        const unsigned long long a_lo = A_ >= 64 ? ~0ull : ((1ull << A_) - 1ull);
        const unsigned long long a_hi = A_ <= 64 ? 0ull : (A_ >= 128 ? ~0ull : ((1ull << (A_ - 64)) - 1ull));
        const unsigned long long adrv_lo = drv_lo & a_lo, adrv_hi = drv_hi & a_hi;
        const int nd = __popcll(drv_lo) + __popcll(drv_hi);
        const int na = __popcll(adrv_lo) + __popcll(adrv_hi);
        const auto kth = [](unsigned long long lo, unsigned long long hi, int k) {  // k-th set bit of (hi:lo), -1 if none
            int slot = -1;
            while (k >= 0) {
                if (lo) {
                    slot = __ffsll((long long)lo) - 1;
                    lo &= lo - 1;
                } else if (hi) {
                    slot = 64 + __ffsll((long long)hi) - 1;
                    hi &= hi - 1;
                } else {
                    return -1;
                }
                --k;
            }
            return slot;
        };
        if (MULTI) {
            // every workgroup of a multi-agent batch is resident at once: the launch lasts as long as its slowest env, and inside
            // an env as long as the wave with the most expensive items -- the waves take the items by ticket, contacts (the
            // expensive ones in a crowd) first
            const int npair = (nd + 1) >> 1;
            for (int guard = 0; guard < npair + na; ++guard) {
                int item = 0;
                if (lane == 0) item = atomicAdd(&l_tk[0], 1);
                item = __builtin_amdgcn_readfirstlane(item);
                if (item < 0 || item >= npair + na) break;
                if (item < na) contacts_vehicle(w, s, c, e, kth(adrv_lo, adrv_hi, item), lane, l_cfl);
                else localize_pair(w, lanes, roads, s, e, kth(drv_lo, drv_hi, 2 * (item - na)), kth(drv_lo, drv_hi, 2 * (item - na) + 1), lane, l_onlane);
            }
        } else if (nd + na <= kWaves) {
            // everything fits one round: one wave per job, the vehicle's data in scalar registers
            for (int item = wave; item < nd + na; item += kWaves) {
                if (item < nd) {
                    if (!(MD_ENV_SKIP & 4)) localize_vehicle(w, lanes, roads, s, e, kth(drv_lo, drv_hi, item), lane, l_onlane);
                } else if (!(MD_ENV_SKIP & 8)) contacts_vehicle(w, s, c, e, kth(adrv_lo, adrv_hi, item - nd), lane, l_cfl);
            }
        } else {
            // many vehicles awake: two localisations per wave (32 lanes each), halving the rounds
            const int npair = (nd + 1) >> 1;
            for (int item = wave; item < npair + na; item += kWaves) {
                if (item < npair) {
                    if (!(MD_ENV_SKIP & 4)) localize_pair(w, lanes, roads, s, e, kth(drv_lo, drv_hi, 2 * item), kth(drv_lo, drv_hi, 2 * item + 1), lane, l_onlane);
                } else if (!(MD_ENV_SKIP & 8))
                    contacts_vehicle(w, s, c, e, kth(adrv_lo, adrv_hi, item - npair), lane, l_cfl);
            }
        }
        __syncthreads();
    } else {
        if (PH & PH_LOCALIZE) {
            for (int j = wave; j < cap; j += kWaves) localize_vehicle(w, lanes, roads, s, e, j, lane);
            __syncthreads();
        }
        MD_STAMP_AT(5);
        if (PH & PH_CONTACTS) {
            for (int j = wave; j < cap; j += kWaves) contacts_vehicle(w, s, c, e, j, lane);
            __syncthreads();
        }
    }
    MD_STAMP_AT(6);
    if (PH & PH_TRAFFIC) {
        for (int j = tid; j < cap; j += kBlock) {
            // Fused step: "drives" comes from the masks built BEFORE the localisation, never from a re-read of the slot's
            // flags: the last wave may already be inside trigger_env below, clearing MD_F_PENDING of slots that did
            // not drive in this step (their l_onlane / l_cfl entries were never written).  Those slots are skipped
            // here by construction, and the read-modify-write of a driving slot's flags cannot collide with the
            // trigger's, which only touches PENDING slots.
            const bool drove = kFused ? ((((j < 64) ? (drv_lo >> j) : (drv_hi >> (j - 64))) & 1ull) != 0ull) : false;
            if (kFused && !drove) continue;
            const int f = s.shape[j].flags;
            if (kFused)  // traffic slots carry no contact flags (l_cfl is written for agents only)
                s.flags[j] = l_onlane[j] | ((f & MD_F_AGENT) ? l_cfl[j] : 0u);
            if (md_drives(f) && !(f & MD_F_AGENT) && !(s.flags[j] & MD_FL_ON_LANE)) {
                s.shape[j].flags = f & ~MD_F_ALIVE;
                if (kFused) l_cfl[j] = kRemovedMark;  // the slot's last write-back (see the dirty-slot write-back)
            }
        }
        // next step's trigger: reads the agents' final lanes and the PENDING slots, which the removal above
        // (slots that drove in this step only) does not touch
        if (plan_ahead && wave == kWaves - 1) trigger_env(lanes, s, c, lane);
        __syncthreads();
        if (RESPAWN) {  // respawn / hybrid: the removed vehicle re-enters on a respawn lane (rare; serial)
            if (tid == 0) md_traffic_respawn_env(&w, lanes, &s, &c, w.env_map[e]);
            __syncthreads();
        }
    }
    MD_STAMP_AT(7);
    if (plan_ahead) {
        // wave 0: the agents' observe chain.  waves 1..: IDM of every driving traffic vehicle for the NEXT step
        // (reads poses / lanes / speeds, writes the traffic slots' action, IDM timer / target lane and PID state:
        // disjoint from what observe writes -- obs, reward, the agent's flags / steps / energy).
        if (wave == 0) {
            if (!(MD_ENV_SKIP & 16))
                for (int a = 0; a < c.agents_per_env; ++a) observe_agent_wave1(lanes, roads, s, c, a, just_reset, lane, l_scratch);
        } else if (!(MD_ENV_SKIP & 2)) {
            // the driving traffic vehicles by RANK, not by slot: dealt by slot (j = A + wave - 1, + kWaves - 1, ...) two of an env's
            // two or three vehicles meet on one wave in every third env while another wave idles
            int rank = 0;
            for (int j0 = 0; j0 < cap; j0 += 64) {
                const int jl = j0 + lane;
                const int fl = (jl < cap) ? s.shape[jl].flags : 0;
                unsigned long long mk = __ballot(jl < cap && md_drives(fl) && !(fl & MD_F_AGENT));
                while (mk) {
                    const int j = j0 + __ffsll((long long)mk) - 1;
                    mk &= mk - 1;
                    if (rank % (kWaves - 1) == wave - 1) idm_vehicle_wave(w, lanes, roads, s, c, w.env_map[e], j, lane, reinterpret_cast<int*>(l_scratch));
                    ++rank;
                }
            }
        }
        MD_STAMP_AT(8);
    } else if (PH & PH_OBSERVE) {
        if (MULTI) {
            const int n_grp = (c.agents_per_env + kObsGroups - 1) / kObsGroups;
            for (int guard = 0; guard < n_grp; ++guard) {
                int gi = 0;
                if (lane == 0) gi = atomicAdd(&l_tk[1], 1);
                gi = __builtin_amdgcn_readfirstlane(gi);
                if (gi < 0 || gi >= n_grp) break;
                observe_agent_wave(lanes, roads, s, c, gi * kObsGroups, just_reset, lane, l_scratch);
            }
        } else {
            for (int a = wave; a < c.agents_per_env; a += kWaves) observe_agent_wave1(lanes, roads, s, c, a, just_reset, lane, l_scratch);
        }
        MD_STAMP_AT(8);
        // lidar only reads shapes; observe writes obs[0:19] / flags / nav / pid -- no barrier needed in between
    }
    if ((PH & PH_LIDAR) && !(PH == PH_ALL && (MD_ENV_SKIP & 1))) {
        if (c.n_beams > 0) phase_lidar(w, s, c, e, tid, kWaves, lidar_out, lidar_stride, lidar_offset, track_det ? l_det : nullptr, (PH == PH_ALL && (MULTI || MD_LEAN_LIDAR_TICKETS)) ? &l_tk[2] : nullptr);
    }

    MD_STAMP_AT(9);
    // ---- write the modified arrays back (coalesced 16-byte stores) ----
    if (!kLidarOnly) {
        __syncthreads();
        MD_STAMP_AT(10);
        constexpr bool respawns = (PH & PH_TRAFFIC) && RESPAWN;  // traffic respawn rewrites a whole slot
        if (kFused && !MULTI && !do_reset) {
            // Single-agent fused step: only slots that drive now (agents, traffic incl. the just triggered /
            // respawned) or were removed this step can differ from what HBM already holds -- props, waiting and
            // dead traffic are never written by any phase.  Writing just those cuts the store traffic ~4x.
            const bool replay = c.traffic_mode == 3;  // replayed slots move without "driving"
            auto dirty = [&](int j) { return replay || md_moves(l_shape[j].flags) || l_cfl[j] == kRemovedMark; };
            for (int i = tid; i < cap * 2; i += kBlock)
                if (dirty(i >> 1)) {
                    st_stream(&reinterpret_cast<uint4*>(gv.shape)[i], reinterpret_cast<const uint4*>(l_shape)[i]);
                    st_stream(&reinterpret_cast<uint4*>(gv.dyn)[i], reinterpret_cast<const uint4*>(l_dyn)[i]);
                    st_stream(&reinterpret_cast<uint4*>(gv.pid)[i], reinterpret_cast<const uint4*>(l_pid)[i]);
                }
            for (int i = tid; i < cap * 4; i += kBlock)
                if (dirty(i >> 2)) st_stream(&reinterpret_cast<uint4*>(gv.nav)[i], reinterpret_cast<const uint4*>(l_nav)[i]);
            for (int j = tid; j < cap; j += kBlock)
                if (dirty(j)) {
                    reinterpret_cast<float2*>(gv.action)[j] = reinterpret_cast<const float2*>(l_action)[j];
                    gv.flags[j] = l_flags[j];
                    if (RESPAWN) gv.final_lane[j] = l_final[j];
                }
            if (track_det)
                for (int j = tid; j < 2 * c.agents_per_env; j += kBlock) gv.detected[j] = l_det[j];
            MD_STAMP_AT(11);
            return;
        }
        if (PH & (PH_RESET | PH_IDM | PH_INTEGRATE | PH_TRAFFIC | PH_OBSERVE | PH_LIFECYCLE)) copy16(gv.shape, l_shape, cap * (int)sizeof(MdShape), tid, kBlock);
        if (MULTI && (PH & (PH_RESET | PH_LIFECYCLE)) && c.random_agent_model && c.is_multi_agent)
            copy16(gv.param, l_param, cap * (int)sizeof(MdParam), tid, kBlock);   // a respawn / reset rewrote vehicle classes
        if ((PH & (PH_RESET | PH_INTEGRATE | PH_LIFECYCLE)) || respawns) copy16(gv.dyn, l_dyn, cap * (int)sizeof(MdDyn), tid, kBlock);
        if ((PH & (PH_RESET | PH_IDM | PH_LOCALIZE | PH_OBSERVE | PH_LIFECYCLE)) || respawns) copy16(gv.nav, l_nav, cap * (int)sizeof(MdNav), tid, kBlock);
        if ((PH & (PH_RESET | PH_IDM | PH_OBSERVE | PH_LIFECYCLE)) || respawns) copy16(gv.pid, l_pid, cap * (int)sizeof(MdPid), tid, kBlock);
        if (((PH & (PH_RESET | PH_LIFECYCLE)) && MULTI) || ((PH & (PH_RESET | PH_TRAFFIC)) && RESPAWN)) {
            for (int j = tid; j < cap; j += kBlock) gv.final_lane[j] = l_final[j];
        }
        for (int j = tid; j < cap; j += kBlock) {
            if ((PH & (PH_RESET | PH_IDM | PH_INTEGRATE | PH_LIFECYCLE)) || respawns) {
                gv.action[2 * j] = l_action[2 * j];
                gv.action[2 * j + 1] = l_action[2 * j + 1];
            }
            if ((PH & (PH_RESET | PH_LOCALIZE | PH_CONTACTS | PH_OBSERVE | PH_LIFECYCLE)) || respawns) gv.flags[j] = l_flags[j];
        }
        if (track_det)
            for (int j = tid; j < 2 * c.agents_per_env; j += kBlock) gv.detected[j] = l_det[j];
        if (do_reset && tid == 0) gv.need_reset[0] = 0;
    }
    MD_STAMP_AT(11);
}

// ------------------------------------------------------------------------------------------------
// wave_step_kernel: the fused step of the SINGLE-AGENT envs with ONE WAVE PER ENVIRONMENT.
//
// env_kernel above gives an env a 4-wave workgroup and orders its phases with workgroup barriers; with one agent and two
// or three driving vehicles per env most of those waves spend most of the step parked at the next barrier (PMC:
// SQ_WAIT_ANY 79 % of the wave-cycles) while holding a wave slot and a share of the LDS, and everything wave-uniform
// (stage-in addressing, masks, flag merges) is executed four times.  Here one wave64 walks through ALL phases of its
// env: the work items that env_kernel deals to waves (localise a vehicle, the agent's contacts, one IDM scan, one
// 64-beam lidar sector) are taken one after the other by the same wave, with the SAME wave-level device functions --
// identical arithmetic, candidate order and tie-breaks, so the results stay bit-identical to the oracle.  What changes
// is the machine mapping:
//   * no workgroup barrier anywhere: phases are ordered by program order inside the wave (LDS operations of one wave
//     execute in order; a compiler fence + wave barrier keeps the compiler from reordering across the lanes' roles);
//   * every resident wave issues useful instructions: 4096 envs = 4096 waves = 4 per SIMD, all of them busy, instead
//     of 1 792 resident workgroups x 4 waves of which ~1.3 per SIMD issue;
//   * a workgroup is kWaveEnvs independent envs (they share nothing but the launch), the LDS image of an env is
//     ~5.4 KB at 24 slots (no routes, no map tables), so LDS never limits the residency.
// Multi-agent envs (40 agents per env) keep env_kernel: there the parallelism inside one env pays.
// ------------------------------------------------------------------------------------------------
#ifndef MD_WAVE_ENVS
#define MD_WAVE_ENVS 4
#endif
constexpr int kWaveEnvs = MD_WAVE_ENVS;   // envs (= waves) per workgroup

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__host__ __device__ inline int wave_env_lds_bytes(int cap, int agents) {
    // shape dyn pid param (32 B) + nav (64 B) per slot, then action (8) flags final onlane cfl (4 each), the wave's
    // scratch (48 floats) and the detected sets (16 B per agent); rounded to 16 B
    const int b = cap * (4 * 32 + 64) + cap * (8 + 4 * 4) + 48 * 4 + 16 * agents;
    return (b + 15) & ~15;
}

// k-th set bit of (hi:lo), -1 if there is none
__device__ __forceinline__ int kth_bit(unsigned long long lo, unsigned long long hi, int k) {
    int slot = -1;
    while (k >= 0) {
        if (lo) {
            slot = __ffsll((long long)lo) - 1;
            lo &= lo - 1;
        } else if (hi) {
            slot = 64 + __ffsll((long long)hi) - 1;
            hi &= hi - 1;
        } else {
            return -1;
        }
        --k;
    }
    return slot;
}

template <bool RESPAWN>
__global__ __launch_bounds__(64 * kWaveEnvs) void wave_step_kernel(MdWorld w, MdState g, MdConfig c, float* lidar_out,
                                                                  int lidar_stride, int lidar_offset) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // blockDim.x / 64 envs per workgroup (<= kWaveEnvs)
    if (e >= c.n_envs) return;   // whole wave
    const int cap = c.cap;
    const int A = c.agents_per_env;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* base = smem + (size_t)(threadIdx.x >> 6) * wave_env_lds_bytes(cap, A);
    MdShape* l_shape = reinterpret_cast<MdShape*>(base);
    MdDyn* l_dyn = reinterpret_cast<MdDyn*>(l_shape + cap);
    MdPid* l_pid = reinterpret_cast<MdPid*>(l_dyn + cap);
    MdParam* l_param = reinterpret_cast<MdParam*>(l_pid + cap);
    MdNav* l_nav = reinterpret_cast<MdNav*>(l_param + cap);
    float* l_action = reinterpret_cast<float*>(l_nav + cap);
    uint32_t* l_flags = reinterpret_cast<uint32_t*>(l_action + 2 * cap);
    int32_t* l_final = reinterpret_cast<int32_t*>(l_flags + cap);
    uint32_t* l_onlane = reinterpret_cast<uint32_t*>(l_final + cap);
    uint32_t* l_cfl = l_onlane + cap;
    float* l_scratch = reinterpret_cast<float*>(l_cfl + cap);
    unsigned long long* l_det = reinterpret_cast<unsigned long long*>(l_scratch + 48);
    const bool track_det = RESPAWN && g.detected != nullptr;

    const MdState gv = md_env_view(&g, &c, e);
    const int reset_flag = gv.need_reset[0];
    MD_STAMP_AT(0);
    const int m = w.env_map[e];
    const MdLane* lanes = w.lanes + w.lane_off[m];
    const MdRoad* roads = w.roads + w.road_off[m];
    const bool fused_act = gv.agent_action != nullptr;

    // ---- stage-in: every global load is issued before the first LDS store (one memory round trip) ----
    {
        const int n32 = cap * 2, n64 = cap * 4;   // 16-byte units
        const uint4* g_shape = reinterpret_cast<const uint4*>(gv.shape);
        const uint4* g_dyn = reinterpret_cast<const uint4*>(gv.dyn);
        const uint4* g_pid = reinterpret_cast<const uint4*>(gv.pid);
        const uint4* g_param = reinterpret_cast<const uint4*>(gv.param);
        const uint4* g_nav = reinterpret_cast<const uint4*>(gv.nav);
        uint4 r_shape, r_dyn, r_pid, r_param, r_nav0, r_nav1;
        float2 r_act;
        uint32_t r_fl;
        int r_fin;
        const bool p32 = lane < n32, pc = lane < cap;
        if (p32) {
            r_shape = ld_stream(&g_shape[lane]);
            r_dyn = ld_stream(&g_dyn[lane]);
            r_pid = ld_stream(&g_pid[lane]);
            r_param = ld_stream(&g_param[lane]);
        }
        if (lane < n64) r_nav0 = ld_stream(&g_nav[lane]);
        if (lane + 64 < n64) r_nav1 = ld_stream(&g_nav[lane + 64]);
        if (pc) {
            r_act = (fused_act && lane < A) ? reinterpret_cast<const float2*>(gv.agent_action)[lane]
                                            : reinterpret_cast<const float2*>(gv.action)[lane];
            r_fl = gv.flags[lane];
            r_fin = gv.final_lane ? gv.final_lane[lane] : 0;
        }
        if (p32) {
            reinterpret_cast<uint4*>(l_shape)[lane] = r_shape;
            reinterpret_cast<uint4*>(l_dyn)[lane] = r_dyn;
            reinterpret_cast<uint4*>(l_pid)[lane] = r_pid;
            reinterpret_cast<uint4*>(l_param)[lane] = r_param;
        }
        if (lane < n64) reinterpret_cast<uint4*>(l_nav)[lane] = r_nav0;
        if (lane + 64 < n64) reinterpret_cast<uint4*>(l_nav)[lane + 64] = r_nav1;
        if (pc) {
            reinterpret_cast<float2*>(l_action)[lane] = r_act;
            l_flags[lane] = r_fl;
            l_final[lane] = r_fin;
        }
        // capacities beyond 32 slots (accident scenes with many props): the tails, in plain loops
        for (int i = lane + 64; i < n32; i += 64) {
            reinterpret_cast<uint4*>(l_shape)[i] = g_shape[i];
            reinterpret_cast<uint4*>(l_dyn)[i] = g_dyn[i];
            reinterpret_cast<uint4*>(l_pid)[i] = g_pid[i];
            reinterpret_cast<uint4*>(l_param)[i] = g_param[i];
        }
        for (int i = lane + 128; i < n64; i += 64) reinterpret_cast<uint4*>(l_nav)[i] = g_nav[i];
        for (int j = lane + 64; j < cap; j += 64) {
            const float* src = (fused_act && j < A) ? gv.agent_action : gv.action;
            l_action[2 * j] = src[2 * j];
            l_action[2 * j + 1] = src[2 * j + 1];
            l_flags[j] = gv.flags[j];
            l_final[j] = gv.final_lane ? gv.final_lane[j] : 0;
        }
        if (track_det)
            for (int j = lane; j < 2 * A; j += 64) l_det[j] = 0ull;
    }
    const bool do_reset = reset_flag != 0;   // wave-uniform
    const int just_reset = do_reset ? 1 : 0;
    if (do_reset) {
        wave_sync();
        const uint4* s0 = reinterpret_cast<const uint4*>(gv.shape0);
        const uint4* d0 = reinterpret_cast<const uint4*>(gv.dyn0);
        const uint4* p0 = reinterpret_cast<const uint4*>(gv.pid0);
        const uint4* n0 = reinterpret_cast<const uint4*>(gv.nav0);
        for (int i = lane; i < cap * 2; i += 64) {
            reinterpret_cast<uint4*>(l_shape)[i] = s0[i];
            reinterpret_cast<uint4*>(l_dyn)[i] = d0[i];
            reinterpret_cast<uint4*>(l_pid)[i] = p0[i];
        }
        for (int i = lane; i < cap * 4; i += 64) reinterpret_cast<uint4*>(l_nav)[i] = n0[i];
        for (int j = lane; j < cap; j += 64) {
            l_action[2 * j] = 0.0f;
            l_action[2 * j + 1] = 0.0f;
            l_flags[j] = 0u;
        }
        if (RESPAWN && (c.traffic_mode == 1 || c.traffic_mode == 2)) {   // respawns rewrote the routes: restore them too
            for (int i = lane; i < cap * MD_ROUTE_LEN; i += 64) {
                gv.route_nodes[i] = gv.route_nodes0[i];
                gv.route_roads[i] = gv.route_roads0[i];
            }
            for (int j = lane; j < cap; j += 64) l_final[j] = gv.final_lane0[j];
        }
    }
    MdState s = gv;   // env-local view whose hot arrays live in this wave's LDS image
    s.shape = l_shape;
    s.dyn = l_dyn;
    s.nav = l_nav;
    s.pid = l_pid;
    s.action = l_action;
    s.flags = l_flags;
    s.param = l_param;
    s.final_lane = l_final;
    wave_sync();
    MD_STAMP_AT(1);

    // agent_policy = IDMPolicy: the agents are planned like the traffic, in the reference's order (decide, then move);
    // otherwise the traffic is planned one step AHEAD, at the end of the step (see env_kernel)
    const bool agent_idm = RESPAWN && c.agent_idm != 0;
    const bool plan_ahead = !agent_idm;
    if (!just_reset && !plan_ahead) {
        trigger_env(lanes, s, c, lane);
        wave_sync();
        for (int j0 = 0; j0 < cap; j0 += 64) {
            const int jj = j0 + lane;
            const unsigned long long mk = __ballot(jj < cap && md_drives(s.shape[jj < cap ? jj : 0].flags));
            if (mk) idm_group_wave(w, lanes, roads, s, c, m, j0, mk, lane, reinterpret_cast<int*>(l_scratch));
        }
        wave_sync();
    }
    MD_STAMP_AT(3);
    // the slots that drive in this step (the integration does not change any slot's flags)
    unsigned long long drv_lo = 0ull, drv_hi = 0ull;
    for (int j0 = 0; j0 < cap; j0 += 64) {
        const int j = j0 + lane;
        const unsigned long long mk = __ballot(j < cap && md_drives(s.shape[j < cap ? j : 0].flags));
        if (j0 == 0) drv_lo = mk;
        else drv_hi = mk;
    }
    if (!just_reset) {
        for (int j = lane; j < cap; j += 64) {
            if (RESPAWN) md_advance_mover(&s, &c, j);
            else md_integrate_mover(&s, &c, j);
        }
        if (!RESPAWN)
            for (int j = lane; j < cap; j += 64) md_walk_mover(&s, &c, j);
        wave_sync();
    }
    MD_STAMP_AT(4);
    // ---- localisation of every driving vehicle, contacts of every driving agent ----
    {
        const unsigned long long a_lo = A >= 64 ? ~0ull : ((1ull << A) - 1ull);
        const unsigned long long a_hi = A <= 64 ? 0ull : (A >= 128 ? ~0ull : ((1ull << (A - 64)) - 1ull));
        const unsigned long long adrv_lo = drv_lo & a_lo, adrv_hi = drv_hi & a_hi;
        const int nd = __popcll(drv_lo) + __popcll(drv_hi);
        const int na = __popcll(adrv_lo) + __popcll(adrv_hi);
        // one vehicle: the whole wave, its data in scalar registers; two: 32 lanes each; more: 16 lanes each, four per pass
        if (nd == 1) localize_vehicle(w, lanes, roads, s, e, kth_bit(drv_lo, drv_hi, 0), lane, l_onlane);
        else if (nd == 2) localize_pair(w, lanes, roads, s, e, kth_bit(drv_lo, drv_hi, 0), kth_bit(drv_lo, drv_hi, 1), lane, l_onlane);
        else
            for (int item = 0; item < nd; item += 4)
                localize_group<16>(w, lanes, roads, s, e, kth_bit(drv_lo, drv_hi, item), kth_bit(drv_lo, drv_hi, item + 1),
                                   kth_bit(drv_lo, drv_hi, item + 2), kth_bit(drv_lo, drv_hi, item + 3), lane, l_onlane);
        for (int k = 0; k < na; ++k) contacts_vehicle(w, s, c, e, kth_bit(adrv_lo, adrv_hi, k), lane, l_cfl);
    }
    wave_sync();
    MD_STAMP_AT(6);
    // ---- flags of the slots that drove; traffic that left every lane is removed ----
    for (int j = lane; j < cap; j += 64) {
        const bool drove = (((j < 64) ? (drv_lo >> j) : (drv_hi >> (j - 64))) & 1ull) != 0ull;
        if (!drove) continue;
        const int f = s.shape[j].flags;
        s.flags[j] = l_onlane[j] | ((f & MD_F_AGENT) ? l_cfl[j] : 0u);
        if (!(f & MD_F_AGENT) && !(s.flags[j] & MD_FL_ON_LANE)) {
            s.shape[j].flags = f & ~MD_F_ALIVE;
            l_cfl[j] = kRemovedMark;
        }
    }
    wave_sync();
    if (plan_ahead) {   // next step's trigger: the agents' final lanes, the PENDING slots
        trigger_env(lanes, s, c, lane);
        wave_sync();
    }
    if (RESPAWN) {   // respawn / hybrid: the removed vehicle re-enters on a respawn lane (rare; serial)
        if (lane == 0) md_traffic_respawn_env(&w, lanes, &s, &c, m);
        wave_sync();
    }
    MD_STAMP_AT(7);
    // ---- next step's traffic decisions, then the agents' observations (disjoint data: either order gives the same) ----
    if (plan_ahead) {
        for (int j0 = 0; j0 < cap; j0 += 64) {
            const int jj = j0 + lane;
            const unsigned long long mk = __ballot(jj < cap && jj >= A && md_drives(s.shape[jj < cap ? jj : 0].flags) &&
                                                   !(s.shape[jj < cap ? jj : 0].flags & MD_F_AGENT));
            if (mk) idm_group_wave(w, lanes, roads, s, c, m, j0, mk, lane, reinterpret_cast<int*>(l_scratch));
            wave_sync();
        }
    }
    MD_STAMP_AT(8);
    for (int a = 0; a < A; ++a) observe_agent_wave1(lanes, roads, s, c, a, just_reset, lane, l_scratch);
    MD_STAMP_AT(9);
    // ---- lidar: the sectors of every agent, one after the other ----
    if (c.n_beams > 0) {
        const int nsec = (c.n_beams + 63) >> 6;
        for (int a = 0; a < A; ++a) {
            float* row = lidar_out + (size_t)(e * A + a) * lidar_stride + lidar_offset;
            for (int sec = 0; sec < nsec; ++sec) lidar_item(w, s, c, a, sec, lane, row, track_det ? l_det + 2 * a : nullptr);
        }
    }
    wave_sync();
    MD_STAMP_AT(10);
    // ---- write-back (16-byte stores) ----
    if (!do_reset) {
        // only slots that move now (agents, traffic incl. the just triggered / respawned, walking participants) or were
        // removed in this step can differ from what HBM already holds
        const bool replay = RESPAWN && c.traffic_mode == 3;   // replayed slots move without "driving"
        auto dirty = [&](int j) { return replay || md_moves(l_shape[j].flags) || l_cfl[j] == kRemovedMark; };
        for (int i = lane; i < cap * 2; i += 64)
            if (dirty(i >> 1)) {
                st_stream(&reinterpret_cast<uint4*>(gv.shape)[i], reinterpret_cast<const uint4*>(l_shape)[i]);
                st_stream(&reinterpret_cast<uint4*>(gv.dyn)[i], reinterpret_cast<const uint4*>(l_dyn)[i]);
                st_stream(&reinterpret_cast<uint4*>(gv.pid)[i], reinterpret_cast<const uint4*>(l_pid)[i]);
            }
        for (int i = lane; i < cap * 4; i += 64)
            if (dirty(i >> 2)) st_stream(&reinterpret_cast<uint4*>(gv.nav)[i], reinterpret_cast<const uint4*>(l_nav)[i]);
        for (int j = lane; j < cap; j += 64)
            if (dirty(j)) {
                reinterpret_cast<float2*>(gv.action)[j] = reinterpret_cast<const float2*>(l_action)[j];
                gv.flags[j] = l_flags[j];
                if (RESPAWN) gv.final_lane[j] = l_final[j];
            }
    } else {
        for (int i = lane; i < cap * 2; i += 64) {
            reinterpret_cast<uint4*>(gv.shape)[i] = reinterpret_cast<const uint4*>(l_shape)[i];
            reinterpret_cast<uint4*>(gv.dyn)[i] = reinterpret_cast<const uint4*>(l_dyn)[i];
            reinterpret_cast<uint4*>(gv.pid)[i] = reinterpret_cast<const uint4*>(l_pid)[i];
        }
        for (int i = lane; i < cap * 4; i += 64) reinterpret_cast<uint4*>(gv.nav)[i] = reinterpret_cast<const uint4*>(l_nav)[i];
        for (int j = lane; j < cap; j += 64) {
            reinterpret_cast<float2*>(gv.action)[j] = reinterpret_cast<const float2*>(l_action)[j];
            gv.flags[j] = l_flags[j];
            if (RESPAWN && gv.final_lane) gv.final_lane[j] = l_final[j];
        }
        if (lane == 0) gv.need_reset[0] = 0;
    }
    if (track_det)
        for (int j = lane; j < 2 * A; j += 64) gv.detected[j] = l_det[j];
    MD_STAMP_AT(11);
}

// ------------------------------------------------------------------------------------------------
// Scenario mode (MdConfig.traffic_mode 4): one ScenarioEnv step per launch, one 4-wave workgroup per scene.
//   stage-in     the scene's mover state -> LDS
//   decide       one WAVE per reactive vehicle (TrajectoryIDMPolicy): projection on its own path with one LANE per
//                polyline segment + wavefront arg-min; arrival; every fifth step (staggered by policy_index) the
//                front search: one lane per candidate object (20 m, any chassis corner inside the path's outline),
//                arg-min of the gaps; PID steering + IDM acceleration on lane 0
//   integrate    one thread per mover (agent + reactive vehicles; replayed ones are kinematic)
//   after_step   wave 0: one lane per track slot -- replay pose of frame k / removal / spawn; the spawns that get a
//                reactive policy take consecutive policy indices through a ballot prefix count (slot order, as the
//                reference's dict order)
//   agent        wave 0 projects the agent on the reference trajectory (lanes = segments) while wave 1 runs its contacts
//   observe      lane 0 of wave 0: state + 22 navigation dims + ScenarioEnv reward / cost / done; lidar sectors on all waves
// The scalar logic is include/md_scenario.h, shared with the oracle (which runs the serial forms).
// ------------------------------------------------------------------------------------------------
// InterpolatingLine.local_coordinates by one wave: lanes = segments (chunks of 64), first minimum wins like np.argmin.
// Every lane first keeps the best of ITS segments (lane, lane + 64, ...: a strict < keeps the earliest), then ONE dense
// wave reduction: the minimum distance by a 6-step butterfly, and among the lanes that hold it the lowest segment index
// (a sparse set, usually one lane: ballot walk).
// EXACT cull of a projection's pieces through MdWorld.poly_ball (a circle per group of MD_POLY_GROUP pieces, lanes = groups, at most
// 64 of them): a piece of the group with the smallest "farthest point of the circle" U is at most U away, so the arg-min lies in a
// group whose circle comes within U; every other group's pieces are farther than the minimum by more than the margin (the radii are
// rounded up by >= 1e-3 m, the test adds 0.02 m + 1e-5 U against the rounding of the distances computed here), so neither the arg-min
// nor a tie can sit in them.  Returns the mask of the groups to evaluate (never empty for n >= 1).
constexpr int kPolyGroup = MD_POLY_GROUP;
static_assert(kPolyGroup == 8, "the lane <-> (group, piece) mapping below packs eight groups of eight pieces into a wave");
__device__ __forceinline__ unsigned long long poly_cull_wave(const float4* balls, int n, float px, float py, int lane_id) {
    const int n_g = (n + kPolyGroup - 1) / kPolyGroup;
    float lb = 3.0e38f, ub = 3.0e38f;
    if (lane_id < n_g) {
        const float4 b = balls[lane_id];
        const float d = md_norm(px - b.x, py - b.y);
        lb = d - b.z;
        ub = d + b.z;
    }
    float U = ub;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) U = md_min(U, __shfl_xor(U, off, 64));
    return __ballot(lane_id < n_g && lb <= U + (0.02f + 1.0e-5f * U));
}
// lane -> the piece it evaluates: piece (lane & 7) of the (lane >> 3)-th lowest group of `cand` (-1: none).  Lanes in ascending order
// hold pieces in ascending order.
__device__ __forceinline__ int cull_piece_of(unsigned long long cand, int n, int lane_id) {
    int grp = -1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int g = cand ? (__ffsll((long long)cand) - 1) : -1;   // wave-uniform
        cand &= cand - 1;
        if ((lane_id >> 3) == q) grp = g;
    }
    const int i = grp * kPolyGroup + (lane_id & 7);
    return (grp >= 0 && i < n) ? i : -1;
}
__device__ __forceinline__ bool poly_can_cull(const float4* balls, int n) { return balls != nullptr && n > 2 * kPolyGroup && n <= 64 * kPolyGroup; }

__device__ __forceinline__ int poly_local_wave(const MdPoly& p, const float4* balls, float px, float py, int lane_id, float* lng, float* lat) {
    // The coordinates come WITHOUT a second trip to memory (arg-min, then md_poly_local_at of that piece): every lane keeps the
    // coordinates w.r.t. the best of its own pieces (the expressions of md_poly_local_at), the winner's are read from its lane.
    float bd = 3.0e38f, bl = 0.0f, bt = 0.0f;
    int bi = 0x7fffffff;
    auto eval = [&](int i) {
        const MdSeg g = p.segs[i];
        const float d = md_seg_dist(&g, px, py);
        if (d < bd) {   // a lane meets its pieces in ascending order: the strict < keeps the earliest of its minima
            const float ddx = px - g.sx, ddy = py - g.sy;
            bd = d;
            bi = i;
            bl = g.cum + (ddx * g.dx + ddy * g.dy);
            bt = ddx * g.dy - ddy * g.dx;
        }
    };
    if (poly_can_cull(balls, p.n)) {
        unsigned long long cand = poly_cull_wave(balls, p.n, px, py, lane_id);
        while (cand) {   // eight groups per pass, ascending
            const int i = cull_piece_of(cand, p.n, lane_id);
#pragma unroll
            for (int q = 0; q < 8; ++q) cand &= cand - 1;
            if (i >= 0) eval(i);
        }
    } else {
        for (int i = lane_id; i < p.n; i += 64) eval(i);
    }
    float m = bd;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = md_min(m, __shfl_xor(m, off, 64));
    const int best = wave_min_i(bi, bd == m, 0x7fffffff);
    const int src = __ffsll((long long)__ballot(bd == m && bi == best)) - 1;   // the one lane that holds it
    *lng = bcast_f(bl, src);
    *lat = bcast_f(bt, src);
    return best;
}

// What a decision needs to know about a slot's route before it touches a piece: staged into LDS with the scene's state (one memory
// round trip for the whole scene) instead of three dependent ones per vehicle (route_n -> poly_off -> aux).  md_route_of + aux.
struct __attribute__((aligned(16))) RouteDesc {
    const MdSeg* segs;
    const float* verts;
    const float4* balls;  // MdWorld.poly_ball of the slot's static polyline; nullptr: none (cut routes, tables not supplied)
    int pad_[2];
    int n;
    int n_verts;          // bit 30: no aux record (end point / outline box are derived by the decision)
    float end_x, end_y;
    float bx0, by0, bx1, by1;
};
constexpr int kDescNoAux = 1 << 30;
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
template <typename T>
__device__ __forceinline__ const T* uni_p(const T* q) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xffffffffull));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32));
    return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
}

// crossing number of the ray from (px, py) over the polygon's edges, lanes = edges: odd = inside (md_point_in_polygon)
__device__ __forceinline__ bool point_in_polygon_wave(const float* xy, int n, float px, float py, int lane_id) {
    int cnt = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane_id;
        cnt += __popcll(__ballot(i < n && md_polygon_edge_crosses(xy, n, i, px, py)));
    }
    return (cnt & 1) != 0;
}

// first segment (ascending) for which `pred` holds, else the last one; pred evaluated by one lane per segment
template <typename F>
__device__ __forceinline__ int poly_first_wave(const MdPoly& p, int lane_id, F pred) {
    for (int i0 = 0; i0 < p.n; i0 += 64) {
        const int i = i0 + lane_id;
        const unsigned long long m = __ballot(i < p.n && pred(p.segs[i]));
        if (m) return i0 + __ffsll((long long)m) - 1;
    }
    return p.n - 1;
}

// The agent on its reference trajectory (md_traj_locate) and the trajectory's length (md_poly_of's) in TWO trips to memory for up to
// 256 pieces: every lane keeps the end longitudinals of its (<= 4) pieces, so the two "first piece that ends beyond" look-ups are
// ballots over registers; then the two pieces they name are read together.
__device__ __forceinline__ void traj_locate_wave(const MdPoly& p, float px, float py, int lane_id, MdTrajLoc* o, float* length) {
    constexpr int kChunks = 4;
    if (p.n > 0 && p.n <= 64 * kChunks) {   // wave-uniform
        float ce[kChunks];
        float bd = 3.0e38f, bl = 0.0f, bt = 0.0f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < kChunks; ++q) {
            const int i = lane_id + 64 * q;
            ce[q] = -3.0e38f;     // never "ends beyond" anything
            if (i < p.n) {
                const MdSeg g = p.segs[i];
                const float d = md_seg_dist(&g, px, py);
                if (d < bd) {
                    const float ddx = px - g.sx, ddy = py - g.sy;
                    bd = d;
                    bi = i;
                    bl = g.cum + (ddx * g.dx + ddy * g.dy);
                    bt = ddx * g.dy - ddy * g.dx;
                }
                ce[q] = g.cum + g.len;
            }
        }
        float m = bd;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = md_min(m, __shfl_xor(m, off, 64));
        const int best = wave_min_i(bi, bd == m, 0x7fffffff);
        const float lng = bcast_f(bl, best & 63);
        o->lng = lng;
        o->lat = bcast_f(bt, best & 63);
        const int il = p.n - 1;
        int ih = il, is = il;
        bool fh = false, fs = false;
        float len = 0.0f;
#pragma unroll
        for (int q = 0; q < kChunks; ++q) {
            const unsigned long long mh = __ballot(ce[q] > lng), ms = __ballot(ce[q] + 0.1f >= lng);
            if (!fh && mh) {
                ih = 64 * q + __ffsll((long long)mh) - 1;
                fh = true;
            }
            if (!fs && ms) {
                is = 64 * q + __ffsll((long long)ms) - 1;
                fs = true;
            }
            if ((il >> 6) == q) len = bcast_f(ce[q], il & 63);
        }
        *length = len;
        const float hh = p.segs[ih].heading;
        const MdSeg gs = p.segs[is];
        o->heading_at = hh;
        o->lat_dx = gs.dy;
        o->lat_dy = -gs.dx;
        return;
    }
    *length = (p.n > 0) ? p.segs[p.n - 1].cum + p.segs[p.n - 1].len : 0.0f;
    poly_local_wave(p, nullptr, px, py, lane_id, &o->lng, &o->lat);
    const float lng = o->lng;
    const int ih = poly_first_wave(p, lane_id, [lng](const MdSeg& g) { return g.cum + g.len > lng; });
    const int is = poly_first_wave(p, lane_id, [lng](const MdSeg& g) { return g.cum + g.len + 0.1f >= lng; });
    o->heading_at = p.segs[ih].heading;
    o->lat_dx = p.segs[is].dy;
    o->lat_dy = -p.segs[is].dx;
}

// Four points against one polygon in one pass over its edges (lanes = edges): bit c of the result = point c is inside
// (odd crossing number, md_point_in_polygon).  `want`: the points worth testing (wave-uniform).
__device__ __forceinline__ int points_in_polygon_wave4(const float* xy, int n, const float* px, const float* py, int want, int lane_id) {
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane_id;
        const bool v = i < n;
        const int ii = v ? i : 0;
        const int j = (ii == 0) ? n - 1 : ii - 1;
        const float xi = xy[2 * ii], yi = xy[2 * ii + 1], xj = xy[2 * j], yj = xy[2 * j + 1];
        // md_polygon_edge_crosses, the same expression for each point
        // the division only where an edge of this chunk straddles the point's y at all (two edges of the whole outline, usually)
#define MD_CROSS(q, cnt)                                                                                       \
        if (want & (1 << q)) {                                                                                 \
            const bool st = v && ((yi > py[q]) != (yj > py[q]));                                               \
            if (__ballot(st)) cnt += __popcll(__ballot(st && (px[q] < (xj - xi) * (py[q] - yi) / (yj - yi) + xi))); \
        }
        MD_CROSS(0, c0)
        MD_CROSS(1, c1)
        MD_CROSS(2, c2)
        MD_CROSS(3, c3)
#undef MD_CROSS
    }
    return (c0 & 1) | ((c1 & 1) << 1) | ((c2 & 1) << 2) | ((c3 & 1) << 3);
}

// TrajectoryIDMPolicy.act of the vehicle in `slot` by one wave (md_tidm_vehicle is the serial form).
// The route's pieces are read ONCE: every lane keeps the end longitudinal and the heading of its (<= 4) pieces in
// registers, so the look-ahead heading after the projection needs no second pass over memory; the end point and the
// outline's bounding box come precomputed (MdWorld.poly_aux); the four chassis corners of a candidate are tested in
// one pass over the outline's edges, and only those inside its bounding box.
// Scratch of the decision stage in LDS (over the route builder's scratch, which the traffic manager uses later in the step).
struct DecideLds {
    unsigned long long* key;   // [cap] (gap bits << 32 | slot) of the front object found so far; kNoFront = none
    float* cur_long;           // [cap] own longitudinal; NaN = arrived (no decision)
    float* heading;            // [cap] heading of the route one metre ahead
    uint32_t* pairs;           // [kPairCap] (vehicle | candidate << 8 | corners-in-box << 16); kNullPair = nothing
    int* ctl;                  // [0] pairs reserved, [1] pair tickets
};
constexpr int kPairCap = 256;
constexpr uint32_t kNullPair = 0xffffffffu;
constexpr unsigned long long kNoFront = ~0ull;
__host__ __device__ constexpr size_t decide_lds_bytes(int cap) { return (size_t)cap * 16 + (size_t)kPairCap * 4 + 16; }
// the scratch both users share: the route builder's [seg_cap + 1] doubles, [seg_cap][2] floats, [seg_cap] ints
__host__ __device__ constexpr size_t sc_scratch_bytes(bool routes, int seg_cap, int cap) {
    const size_t a = routes ? (size_t)seg_cap * 20 + 8 : 0, b = decide_lds_bytes(cap);
    return ((a > b ? a : b) + 15) & ~(size_t)15;
}

// One (speed-control vehicle, near mover) pair of get_find_front_back_objs_single_lane (md_tidm_front_gap), by one wave: the chassis
// corners inside the outline's bounding box against the outline (lanes = polygon edges, all corners in one pass), then the
// projection on the route with lanes = pieces.  A mover that counts enters the vehicle's key by an LDS atomic min: the smallest gap,
// among equal gaps the lowest slot -- what the serial search's ascending order and strict < leave.
__device__ __forceinline__ void tidm_pair_wave(const MdState& s, int slot, int j, int want, int lane_id, const RouteDesc* l_desc,
                                               const DecideLds& dl) {
    const RouteDesc rd = l_desc[slot];
    MdPoly route;
    route.segs = uni_p(rd.segs);
    route.n = uni_i(rd.n);
    route.length = 0.0f;
    const float* pv = uni_p(rd.verts);
    const int n_v = uni_i(rd.n_verts) & ~kDescNoAux;
    const float4* balls = uni_p(rd.balls);
    const MdShape o = s.shape[j];   // wave-uniform
    const float ex = o.c * o.hl, ey = o.s * o.hl, fx = -o.s * o.hw, fy = o.c * o.hw;
    const float qx[4] = {o.cx + ex + fx, o.cx + ex - fx, o.cx - ex - fx, o.cx - ex + fx};
    const float qy[4] = {o.cy + ey + fy, o.cy + ey - fy, o.cy - ey - fy, o.cy - ey + fy};
    if (points_in_polygon_wave4(pv, n_v, qx, qy, want, lane_id) == 0) return;
    float lg, lt;
    poly_local_wave(route, balls, o.cx, o.cy, lane_id, &lg, &lt);
    const float gap = lg - dl.cur_long[slot];
    if (lane_id == 0 && gap > 0.0f && gap < MD_TIDM_MAX_DIST)
        atomicMin(&dl.key[slot], ((unsigned long long)__float_as_uint(gap) << 32) | (unsigned)j);
}

// TrajectoryIDMPolicy.act of the vehicle in `slot`, everything BEFORE the front search's candidates are looked at, by one wave
// (md_tidm_vehicle is the serial form): arrival, own projection, the heading ahead; a vehicle due for speed control leaves its
// (vehicle, near mover) pairs on the scene's pair list, which ALL waves then work off (tidm_pair_wave) -- a crowded vehicle's
// search was the tail of the stage when one wave walked its movers alone.
__device__ __forceinline__ void tidm_prepare_wave(const MdWorld& w, const MdState& s, const MdConfig& c, int e, int slot, int k,
                                  int lane_id, const RouteDesc* l_desc, const DecideLds& dl) {
#ifdef MD_STAMP
    bool st_ = lane_id == 0;   // diagnostic: the scene's first reactive vehicle, apart for its speed-control steps (slots 16.. / 24..)
    for (int q = c.agents_per_env; q < slot; ++q) st_ = st_ && !(s.nav[q].ck0 == MD_SC_IDM && md_present(s.shape[q].flags));
    const int so_ = ((k % MD_TIDM_BATCH) == s.nav[slot].timer) ? 8 : 0;
#endif
    MD_FINE_STAMP(st_, so_ + 0);
    // the slot's static polyline or the route cut at its spawn frame (md_route_of), end point and outline box (aux): the record the
    // stage-in left in LDS -- wave-uniform, kept in scalar registers
    const RouteDesc rd = l_desc[slot];
    MdPoly route;
    route.segs = uni_p(rd.segs);
    route.n = uni_i(rd.n);
    route.length = 0.0f;   // only the fallback below needs it: a dependent load of the last piece
    const int nvf = uni_i(rd.n_verts);
    const float4* balls = uni_p(rd.balls);
    const float px = s.shape[slot].cx, py = s.shape[slot].cy;
    float end_x, end_y, bx0 = -3.0e38f, by0 = -3.0e38f, bx1 = 3.0e38f, by1 = 3.0e38f;
    if (!(nvf & kDescNoAux)) {
        end_x = uni_f(rd.end_x);
        end_y = uni_f(rd.end_y);
        bx0 = uni_f(rd.bx0);
        by0 = uni_f(rd.by0);
        bx1 = uni_f(rd.bx1);
        by1 = uni_f(rd.by1);
    } else {
        route.length = (route.n > 0) ? route.segs[route.n - 1].cum + route.segs[route.n - 1].len : 0.0f;
        const float length = route.length;
        const int ie = poly_first_wave(route, lane_id, [length](const MdSeg& g) { return g.cum + g.len + 0.1f >= length; });
        const MdSeg ge = route.segs[ie];
        end_x = ge.sx + (length - ge.cum) * ge.dx;
        end_y = ge.sy + (length - ge.cum) * ge.dy;
    }
    if (md_norm(px - end_x, py - end_y) < MD_TIDM_DEST_RADIUS) {
        if (lane_id == 0) {
            s.nav[slot].ck0 = MD_SC_ARRIVED;   // no action this step: the vehicle rolls on with its previous one
            dl.cur_long[slot] = __int_as_float(0x7fc00000);
        }
        return;
    }
    MD_FINE_STAMP(st_, so_ + 1);
    const int do_speed_control = (k % MD_TIDM_BATCH) == s.nav[slot].timer;
    // ---- projection on the own route; the pieces' end longitudinals / headings stay in registers ----
    constexpr int kChunks = 4;
    // (1) with the route's group circles (poly_cull_wave) and at most eight groups left: ONE pass, a lane per surviving piece -- the
    // usual case, ~20 of 100 pieces; (2) up to 256 pieces: all of them, (<= 4) per lane; (3) anything else: the generic loops
    unsigned long long cand = 0ull;
    if (poly_can_cull(balls, route.n)) cand = poly_cull_wave(balls, route.n, px, py, lane_id);
    const bool culled = cand != 0ull && __popcll(cand) <= 8;   // wave-uniform
    const bool small = !culled && route.n <= 64 * kChunks;      // wave-uniform
    float ce[kChunks], hd[kChunks];
    int best;
    float cur_long;
    int my_i = -1;             // (1): this lane's piece, its end longitudinal and heading
    float my_ce = -3.0e38f, my_hd = 0.0f;
    if (culled) {
        my_i = cull_piece_of(cand, route.n, lane_id);
        float bd = 3.0e38f, bl = 0.0f;
        if (my_i >= 0) {
            const MdSeg g = route.segs[my_i];
            bd = md_seg_dist(&g, px, py);
            bl = g.cum + ((px - g.sx) * g.dx + (py - g.sy) * g.dy);   // md_poly_local_at w.r.t. this piece
            my_ce = g.cum + g.len;
            my_hd = g.heading;
        }
        float m = bd;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = md_min(m, __shfl_xor(m, off, 64));
        const unsigned long long mm = __ballot(bd == m);   // lanes ascend with the pieces: the lowest lane = the lowest index
        const int src = __ffsll((long long)mm) - 1;
        best = bcast_i(my_i, src);
        cur_long = bcast_f(bl, src);
    } else if (small) {
        float bd = 3.0e38f, bl = 0.0f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < kChunks; ++q) {
            const int i = lane_id + 64 * q;
            ce[q] = -3.0e38f;     // never "ends beyond" anything
            hd[q] = 0.0f;
            if (i < route.n) {
                const MdSeg g = route.segs[i];
                const float d = md_seg_dist(&g, px, py);
                if (d < bd) {
                    bd = d;
                    bi = i;
                    bl = g.cum + ((px - g.sx) * g.dx + (py - g.sy) * g.dy);   // md_poly_local_at w.r.t. this piece
                }
                ce[q] = g.cum + g.len;
                hd[q] = g.heading;
            }
        }
        float m = bd;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = md_min(m, __shfl_xor(m, off, 64));
        best = wave_min_i(bi, bd == m, 0x7fffffff);
        cur_long = bcast_f(bl, best & 63);   // the lane of piece `best`; its own best is that piece (strict <)
    } else {
        float tmp;
        best = poly_local_wave(route, balls, px, py, lane_id, &cur_long, &tmp);
    }
    MD_FINE_STAMP(st_, so_ + 2);
    if (do_speed_control) {
        // md_tidm_front_gap's filters that need no route: lanes = movers -- present, within 20 m, and a chassis corner inside the
        // outline's bounding box (a point outside it is outside the outline).  The survivors go on the pair list; when the list
        // is full (never in practice: kPairCap pairs per scene) this wave works its pairs off itself, after publishing cur_long.
        if (lane_id == 0) dl.cur_long[slot] = cur_long;
        wave_sync();
        for (int j0 = 0; j0 < c.cap; j0 += 64) {
            const int jl = j0 + lane_id;
            int want = 0;
            if (jl < c.cap && jl != slot) {
                const MdShape o = s.shape[jl];
                if (md_present(o.flags) && !(md_norm(o.cx - px, o.cy - py) > MD_TIDM_MAX_DIST)) {
                    const float ex = o.c * o.hl, ey = o.s * o.hl, fx = -o.s * o.hw, fy = o.c * o.hw;
                    const float qx[4] = {o.cx + ex + fx, o.cx + ex - fx, o.cx - ex - fx, o.cx - ex + fx};
                    const float qy[4] = {o.cy + ey + fy, o.cy + ey - fy, o.cy - ey - fy, o.cy - ey + fy};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (!(qx[q] < bx0 || qx[q] > bx1 || qy[q] < by0 || qy[q] > by1)) want |= 1 << q;
                }
            }
            unsigned long long mk = __ballot(want != 0);
            if (mk == 0) continue;
            const int cnt = __popcll(mk);
            int r0 = 0;
            if (lane_id == 0) r0 = atomicAdd(&dl.ctl[0], cnt);
            r0 = uni_i(r0);
            if (r0 + cnt <= kPairCap) {
                if (want != 0) dl.pairs[r0 + __popcll(mk & ((1ull << lane_id) - 1ull))] = (uint32_t)slot | ((uint32_t)jl << 8) | ((uint32_t)want << 16);
            } else {
                for (int i = r0 + lane_id; i < kPairCap; i += 64) dl.pairs[i] = kNullPair;   // the part of the reservation inside the list
                while (mk) {
                    const int b = __ffsll((long long)mk) - 1;
                    mk &= mk - 1;
                    tidm_pair_wave(s, slot, j0 + b, uni_i(__shfl(want, b, 64)), lane_id, l_desc, dl);
                }
            }
        }
    }
    MD_FINE_STAMP(st_, so_ + 3);
    // heading of the route one metre ahead: the first piece that ends beyond it (md_poly_seg_heading), else the last one
    const float ahead = cur_long + 1.0f;
    float lane_heading = 0.0f;
    bool have_heading = false;
    if (culled) {
        // the first piece that ends beyond `ahead`, among the pieces in the lanes: it is THE first one when the piece before it is in
        // the lanes too (and does not end beyond) or there is none before it -- every piece but a route's last is longer than 1 m,
        // so the end longitudinals ascend.  Otherwise (the look-ahead leaves the surviving groups) the generic search below.
        const unsigned long long mf = __ballot(my_i >= 0 && my_ce > ahead);
        if (mf) {
            const int src = __ffsll((long long)mf) - 1;
            const int ih = bcast_i(my_i, src);
            const int prev = (src > 0) ? bcast_i(my_i, src - 1) : -2;
            if (ih == 0 || prev == ih - 1) {
                lane_heading = bcast_f(my_hd, src);
                have_heading = true;
            }
        }
    }
    if (have_heading) {
    } else if (small) {
        int ih = route.n - 1;
        bool found = false;
#pragma unroll
        for (int q = 0; q < kChunks; ++q) {
            const unsigned long long m = __ballot(ce[q] > ahead);
            if (!found && m) {
                ih = 64 * q + __ffsll((long long)m) - 1;
                found = true;
            }
        }
        lane_heading = 0.0f;
#pragma unroll
        for (int q = 0; q < kChunks; ++q)
            if ((ih >> 6) == q) lane_heading = bcast_f(hd[q], ih & 63);
    } else {
        const int ih = poly_first_wave(route, lane_id, [ahead](const MdSeg& g) { return g.cum + g.len > ahead; });
        lane_heading = route.segs[ih].heading;
    }
    MD_FINE_STAMP(st_, so_ + 4);
    if (lane_id == 0) {
        dl.cur_long[slot] = cur_long;
        dl.heading[slot] = lane_heading;
    }
    MD_FINE_STAMP(st_, so_ + 5);
}

// The route cut at a spawn frame, built by one wave from the positions staged in LDS (md_build_route is the serial form; the element
// functions are the same, so the results are bit-identical): the chain of kept points with lanes = candidate points (one ballot per
// piece), the pieces with lanes = pieces (dropped ones squeezed out by ballot prefix counts), the running length on lane 0 (a sum in
// a fixed order), the outline with lanes = vertices.  A few per scene and episode; ~3 us instead of ~40 on one lane.
//   l_link [seg_cap] ints, l_len [seg_cap + 1] doubles: LDS scratch.  rn = the slot's route_n[4].
__device__ __forceinline__ void build_route_wave(int32_t* rn, MdSeg* segs, int seg_cap, float* verts, int vert_cap, float* aux,
                                 const float* l_pts, int n_pts, int* l_link, double* l_len, int lane) {
    int nl = 0, i = 0;
    while (i < n_pts - 1 && nl < seg_cap) {
        int j = n_pts - 1;
        for (int q0 = i + 1; q0 < n_pts; q0 += 64) {
            const int q = q0 + lane;
            const unsigned long long m = __ballot(q < n_pts && md_route_far(l_pts, 2, i, q));
            if (m) {
                j = q0 + __ffsll((long long)m) - 1;
                break;
            }
        }
        if (lane == 0) l_link[nl] = i | (j << 16);
        ++nl;
        i = j;
    }
    wave_sync();
    int ns = 0;
    for (int p0 = 0; p0 < nl; p0 += 64) {
        const int p = p0 + lane;
        MdSeg g;
        double L = -1.0;
        if (p < nl) {
            const int lk = l_link[p];
            L = md_route_piece(l_pts, 2, lk & 0xffff, lk >> 16, &g);
        }
        const bool keep = !(L < 0.0);
        const unsigned long long m = __ballot(keep);
        const int at = ns + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) {
            segs[at] = g;
            l_len[at] = L;
        }
        ns += __popcll(m);
    }
    __threadfence_block();
    wave_sync();
    const int never_moved = ns == 0;
    if (lane == 0) {
        if (never_moved) {
            md_route_still_piece(l_pts, &segs[0]);
            l_len[seg_cap] = 0.1;
        } else {
            double cum = 0.0;
            for (int p = 0; p < ns; ++p) {
                segs[p].cum = (float)cum;
                cum += l_len[p];
            }
            l_len[seg_cap] = cum;
        }
    }
    if (never_moved) ns = 1;
    __threadfence_block();
    wave_sync();
    const int n_long = md_route_n_long(l_len[seg_cap]);
    const int nv = 2 * (n_long + 2);
    if (nv > vert_cap) {   // (the host sizes the buffers by the longest run: never) the slot keeps its static polyline
        if (lane == 0) rn[0] = rn[1] = rn[3] = 0;
        return;
    }
    float bx0 = 3.0e38f, by0 = 3.0e38f, bx1 = -3.0e38f, by1 = -3.0e38f;
    for (int o0 = 0; o0 < nv; o0 += 64) {
        const int o = o0 + lane;
        if (o < nv) {
            float fx, fy;
            md_route_outline_vertex(segs, ns, never_moved, n_long, o, &fx, &fy);
            verts[2 * o] = fx;
            verts[2 * o + 1] = fy;
            bx0 = md_min(bx0, fx);
            by0 = md_min(by0, fy);
            bx1 = md_max(bx1, fx);
            by1 = md_max(by1, fy);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        bx0 = md_min(bx0, __shfl_xor(bx0, off, 64));
        by0 = md_min(by0, __shfl_xor(by0, off, 64));
        bx1 = md_max(bx1, __shfl_xor(bx1, off, 64));
        by1 = md_max(by1, __shfl_xor(by1, off, 64));
    }
    if (lane == 0) {
        md_route_end_point(segs, ns, aux);
        aux[2] = bx0;
        aux[3] = by0;
        aux[4] = bx1;
        aux[5] = by1;
        aux[6] = aux[7] = 0.0f;
        rn[0] = ns;
        rn[1] = nv;
        rn[3] = 0;
    }
}

// Register budget of the scenario kernel: 8 waves per SIMD (64 VGPRs, one spilled) -- 2048 scenes = 256 CUs x 8 workgroups
// are then resident at once, one round instead of two (measured 148 vs 173 us at the compiler's own choice)
// Diagnostic builds only (tools/ab/sc_knockout.sh): stages left out to read their marginal cost off the launch time.  0 in the product.
#ifndef MD_SC_SKIP
#define MD_SC_SKIP 0
#endif
#ifndef MD_SC_WAVES_EU
#define MD_SC_WAVES_EU 8
#endif
__global__ __launch_bounds__(256)
#if MD_SC_WAVES_EU
__attribute__((amdgpu_waves_per_eu(MD_SC_WAVES_EU, MD_SC_WAVES_EU)))
#endif
void scenario_step_kernel(MdWorld w, MdState g, MdConfig c, float* lidar_out, int lidar_stride,
                                                           int lidar_offset) {
    constexpr int kBlock = 256, kWaves = 4;
    const int e = blockIdx.x;
    if (e >= c.n_envs) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int cap = c.cap, A = c.agents_per_env;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    MdShape* l_shape = reinterpret_cast<MdShape*>(smem);
    MdDyn* l_dyn = reinterpret_cast<MdDyn*>(l_shape + cap);
    MdPid* l_pid = reinterpret_cast<MdPid*>(l_dyn + cap);
    MdParam* l_param = reinterpret_cast<MdParam*>(l_pid + cap);
    MdNav* l_nav = reinterpret_cast<MdNav*>(l_param + cap);
    float* l_action = reinterpret_cast<float*>(l_nav + cap);
    uint32_t* l_flags = reinterpret_cast<uint32_t*>(l_action + 2 * cap);
    uint32_t* l_cfl = l_flags + cap;
    MdTrajLoc* l_loc = reinterpret_cast<MdTrajLoc*>(l_cfl + ((cap + 3) & ~3));   // [A]
    int* l_count = reinterpret_cast<int*>(l_loc + A);
    int* l_dbest = l_count + 4;   // [A][n_side + n_lane_line] detector fractions (bit patterns), when md_step runs the detectors
    const bool fused_det = (w.side_beam_cs != nullptr && c.n_side > 0) || (w.ll_beam_cs != nullptr && c.n_lane_line > 0);
    const int n_det = (w.side_beam_cs ? c.n_side : 0) + (w.ll_beam_cs ? c.n_lane_line : 0);
    // [route_seg_cap][2]: the positions a route cut at a spawn frame is built from (behind the detectors' share, as md_step sizes it)
    // and the builder's scratch: [seg_cap + 1] doubles, [seg_cap] ints
    double* l_len = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(l_count + 4 + (A + 2) * (c.n_side + c.n_lane_line) +
                                                                          ((c.n_side + c.n_lane_line) > 0 ? 2 * kDetPairs : 0)) + 7) & ~(uintptr_t)7);
    float* l_pts = reinterpret_cast<float*>(l_len + c.route_seg_cap + 1);
    int* l_link = reinterpret_cast<int*>(l_pts + 2 * c.route_seg_cap);
    // the decision stage's scratch shares the builder's (the traffic manager builds routes later in the step)
    DecideLds dl;
    dl.key = reinterpret_cast<unsigned long long*>(l_len);
    dl.cur_long = reinterpret_cast<float*>(dl.key + cap);
    dl.heading = dl.cur_long + cap;
    dl.pairs = reinterpret_cast<uint32_t*>(dl.heading + cap);
    dl.ctl = reinterpret_cast<int*>(dl.pairs + kPairCap);
    unsigned char* l_scratch_end = reinterpret_cast<unsigned char*>(l_len) + sc_scratch_bytes(g.route_n != nullptr, c.route_seg_cap, cap);
    int32_t* l_rn = reinterpret_cast<int32_t*>(l_scratch_end);   // [cap][4]: route_n of the scene (the decisions read it first: not a global round trip)
    // [cap] the movers as the agent's contact test sees them: after the integration, BEFORE the traffic manager's after_step
    MdShape* l_shape_ct = reinterpret_cast<MdShape*>((reinterpret_cast<uintptr_t>(g.route_n != nullptr ? (void*)(l_rn + 4 * cap) : (void*)l_scratch_end) + 15) & ~(uintptr_t)15);
    RouteDesc* l_desc = reinterpret_cast<RouteDesc*>(l_shape_ct + cap);   // [cap] what the decisions know about each slot's route

    MD_STAMP_AT(0);
    const MdState gv = md_env_view(&g, &c, e);
    const int reset_flag = gv.need_reset[0];
    const bool fused_act = gv.agent_action != nullptr;
    const bool do_reset = reset_flag != 0;
    {
        // ONE memory round trip: every global load of the live state is issued before the first LDS store, and nothing waits for
        // the reset flag (a chain of copy loops waits for each loop's loads in turn: ~15 k cycles per scene); only a scene that
        // resets re-stages from the snapshot below.  16-byte units: shape / dyn / pid / param cap * 2 each, nav cap * 4.
        const int n32 = cap * 2, n64 = cap * 4;
        const uint4* g_shape = reinterpret_cast<const uint4*>(gv.shape);
        const uint4* g_dyn = reinterpret_cast<const uint4*>(gv.dyn);
        const uint4* g_pid = reinterpret_cast<const uint4*>(gv.pid);
        const uint4* g_param = reinterpret_cast<const uint4*>(gv.param);
        const uint4* g_nav = reinterpret_cast<const uint4*>(gv.nav);
        uint4 r_shape, r_dyn, r_pid, r_param, r_nav0, r_nav1;
        float2 r_act;
        uint32_t r_fl;
        int po0 = 0, po1 = 0, pv0 = 0, pv1 = 0, pb0 = 0, rn0 = 0, rn1 = 0, rn2 = 0, rn3 = 0, r_cnt = 0;
        const bool has_balls = w.poly_ball != nullptr && w.poly_ball_off != nullptr;
        float ax[6];
        const bool p32 = tid < n32, pc = tid < cap;
        if (p32) {
            r_shape = g_shape[tid];
            r_dyn = g_dyn[tid];
            r_pid = g_pid[tid];
            r_param = g_param[tid];
        }
        if (tid < n64) r_nav0 = g_nav[tid];
        if (tid + kBlock < n64) r_nav1 = g_nav[tid + kBlock];
        if (pc) {
            r_act = (fused_act && tid < A) ? reinterpret_cast<const float2*>(gv.agent_action)[tid] : reinterpret_cast<const float2*>(gv.action)[tid];
            r_fl = gv.flags[tid];
            // the slot's route record (md_route_of + MdWorld.poly_aux): every address is known here
            const size_t ng = (size_t)e * cap + tid;
            po0 = w.poly_off[ng];
            po1 = w.poly_off[ng + 1];
            pv0 = w.polyv_off[ng];
            pv1 = w.polyv_off[ng + 1];
            if (has_balls) pb0 = w.poly_ball_off[ng];
            if (w.poly_aux) {
#pragma unroll
                for (int q = 0; q < 6; ++q) ax[q] = w.poly_aux[8 * ng + q];
            }
            if (gv.route_n) {
                rn0 = gv.route_n[4 * tid];
                rn1 = gv.route_n[4 * tid + 1];
                rn2 = gv.route_n[4 * tid + 2];
                rn3 = gv.route_n[4 * tid + 3];
            }
        }
        if (tid == 0) r_cnt = gv.next_agent_id[0];
        if (p32) {
            reinterpret_cast<uint4*>(l_shape)[tid] = r_shape;
            reinterpret_cast<uint4*>(l_dyn)[tid] = r_dyn;
            reinterpret_cast<uint4*>(l_pid)[tid] = r_pid;
            reinterpret_cast<uint4*>(l_param)[tid] = r_param;
        }
        if (tid < n64) reinterpret_cast<uint4*>(l_nav)[tid] = r_nav0;
        if (tid + kBlock < n64) reinterpret_cast<uint4*>(l_nav)[tid + kBlock] = r_nav1;
        if (pc) {
            reinterpret_cast<float2*>(l_action)[tid] = r_act;
            l_flags[tid] = r_fl;
            if (do_reset) rn0 = rn1 = rn2 = rn3 = 0;   // a reset leaves no cut routes
            if (gv.route_n) {
                l_rn[4 * tid] = rn0;
                l_rn[4 * tid + 1] = rn1;
                l_rn[4 * tid + 2] = rn2;
                l_rn[4 * tid + 3] = rn3;
            }
            RouteDesc d;
            d.pad_[0] = d.pad_[1] = 0;
            if (rn0 > 0 && tid >= A) {   // a route cut at a spawn frame (rare; never an agent's: its slot keeps the reference trajectory): its record lives in the state, one more trip for this lane
                d.segs = gv.route_segs + (size_t)tid * c.route_seg_cap;
                d.verts = gv.route_verts + 2 * (size_t)tid * c.route_vert_cap;
                d.balls = nullptr;
                d.n = rn0;
                d.n_verts = rn1;
                const float* ra = gv.route_aux + 8 * (size_t)tid;
                d.end_x = ra[0];
                d.end_y = ra[1];
                d.bx0 = ra[2];
                d.by0 = ra[3];
                d.bx1 = ra[4];
                d.by1 = ra[5];
            } else {
                d.segs = w.segs + po0;
                d.verts = w.polyv + 2 * (size_t)pv0;
                d.balls = has_balls ? reinterpret_cast<const float4*>(w.poly_ball) + pb0 : nullptr;
                d.n = po1 - po0;
                d.n_verts = (pv1 - pv0) | (w.poly_aux ? 0 : kDescNoAux);
                d.end_x = w.poly_aux ? ax[0] : 0.0f;
                d.end_y = w.poly_aux ? ax[1] : 0.0f;
                d.bx0 = w.poly_aux ? ax[2] : 0.0f;
                d.by0 = w.poly_aux ? ax[3] : 0.0f;
                d.bx1 = w.poly_aux ? ax[4] : 0.0f;
                d.by1 = w.poly_aux ? ax[5] : 0.0f;
            }
            l_desc[tid] = d;
        }
        if (tid == 0) *l_count = do_reset ? 0 : r_cnt;   // idm_policy_count
    }
    if (do_reset) {   // block-uniform, rare: the snapshot over what was just staged (same threads wrote the same words: no barrier needed
                      // between a thread's own stores, and the barrier below orders everything before the first read)
        __syncthreads();
        copy16(l_shape, gv.shape0, cap * (int)sizeof(MdShape), tid, kBlock);
        copy16(l_dyn, gv.dyn0, cap * (int)sizeof(MdDyn), tid, kBlock);
        copy16(l_pid, gv.pid0, cap * (int)sizeof(MdPid), tid, kBlock);
        copy16(l_nav, gv.nav0, cap * (int)sizeof(MdNav), tid, kBlock);
        for (int j = tid; j < cap; j += kBlock) {
            l_action[2 * j] = 0.0f;
            l_action[2 * j + 1] = 0.0f;
            l_flags[j] = 0u;
        }
    }
    float* l_beams = reinterpret_cast<float*>(l_dbest + A * n_det);   // [n_det][2]: the beam tables, read n_beams times per quad
    if (fused_det) {
        const int ns = w.side_beam_cs ? c.n_side : 0;
        for (int it = tid; it < A * n_det; it += kBlock) l_dbest[it] = __float_as_int(1.0f);
        for (int it = tid; it < 2 * n_det; it += kBlock)
            l_beams[it] = (it < 2 * ns) ? w.side_beam_cs[it] : w.ll_beam_cs[it - 2 * ns];
    }
    MdState s = gv;
    if (gv.route_n) s.route_n = l_rn;
    s.shape = l_shape;
    s.dyn = l_dyn;
    s.nav = l_nav;
    s.pid = l_pid;
    s.action = l_action;
    s.flags = l_flags;
    s.param = l_param;
    s.next_agent_id = l_count;
    __syncthreads();
    MD_STAMP_AT(1);
    const int just_reset = do_reset ? 1 : 0;
    const int k = just_reset ? 0 : s.nav[0].steps + 1;   // engine.episode_step of this step
    constexpr int kSkip = MD_SC_SKIP;
    if (!just_reset) {
        // Work list of the reactive vehicles, those due for speed control in this step FIRST (their front search makes them
        // ~2.5x as expensive as the others); the waves then take vehicles off the list one by one through an LDS counter,
        // so a wave that drew a cheap vehicle moves on instead of waiting at the barrier (every wave leaves the loop as
        // soon as the counter passes the list's end: no waiting inside).  The decisions are independent of each other,
        // so the order changes nothing in the results.
        int* l_list = reinterpret_cast<int*>(l_cfl);   // free until the contacts stage writes its flags
        int* l_ctr = l_count + 1;
        int* l_n = l_count + 2;
        if (wave == 0) {
            int n = 0;
            for (int pass = 0; pass < 2; ++pass)
                for (int j0 = A; j0 < cap; j0 += 64) {
                    const int j = j0 + lane;
                    bool take = false;
                    if (j < cap && s.nav[j].ck0 == MD_SC_IDM && md_present(s.shape[j].flags)) {
                        const bool sc = (k % MD_TIDM_BATCH) == s.nav[j].timer;
                        take = (pass == 0) ? sc : !sc;
                    }
                    const unsigned long long m = __ballot(take);
                    if (take) l_list[n + __popcll(m & ((1ull << lane) - 1ull))] = j;
                    n += __popcll(m);
                }
            if (lane == 0) {
                *l_n = n;
                *l_ctr = 0;
            }
        }
        for (int j = tid; j < cap; j += kBlock) dl.key[j] = kNoFront;
        if (tid == 64) dl.ctl[0] = dl.ctl[1] = 0;
        __syncthreads();
        const int n_list = __builtin_amdgcn_readfirstlane(*l_n);
        // (A) per vehicle: arrival, own projection, heading ahead, the speed-control vehicles' candidate pairs
        for (int guard = 0; guard < cap; ++guard) {   // at most cap tickets per wave: the loop ends whatever the counter holds
            int i = 0;
            if (lane == 0) i = atomicAdd(l_ctr, 1);
            i = __builtin_amdgcn_readfirstlane(i);     // lane 0's ticket, in a scalar register
            if (i < 0 || i >= n_list) break;
            const int slot = __builtin_amdgcn_readfirstlane(l_list[i]);
            if (slot < A || slot >= cap) break;        // never index with anything but a mover slot
            if (kSkip & 32) {
                if (lane == 0) dl.cur_long[slot] = __int_as_float(0x7fc00000);
                continue;
            }
            tidm_prepare_wave(w, s, c, e, slot, k, lane, l_desc, dl);
        }
        __syncthreads();
        // (B) the pairs, one per ticket
        const int n_pairs = min(__builtin_amdgcn_readfirstlane(dl.ctl[0]), kPairCap);
        for (int guard = 0; guard < kPairCap; ++guard) {
            int i = 0;
            if (lane == 0) i = atomicAdd(&dl.ctl[1], 1);
            i = __builtin_amdgcn_readfirstlane(i);
            if (i < 0 || i >= n_pairs) break;
            const uint32_t pr = (uint32_t)__builtin_amdgcn_readfirstlane((int)dl.pairs[i]);
            if (pr == kNullPair || (kSkip & 16)) continue;
            const int slot = pr & 0xff, j = (pr >> 8) & 0xff, want = (pr >> 16) & 0xf;
            if (slot < A || slot >= cap || j >= cap) break;
            tidm_pair_wave(s, slot, j, want, lane, l_desc, dl);
        }
        __syncthreads();
        // (C) steering + acceleration (md_tidm_decide): one lane per vehicle
        for (int i = tid; i < n_list; i += kBlock) {
            const int slot = l_list[i];
            const float cl = dl.cur_long[slot];
            if (slot >= A && slot < cap && cl == cl) {   // NaN: arrived
                const unsigned long long key = dl.key[slot];
                const int front = (key == kNoFront) ? -1 : (int)(key & 0xffffffffull);
                const float front_dist = (key == kNoFront) ? MD_TIDM_MAX_DIST : __uint_as_float((unsigned)(key >> 32));
                const int do_speed_control = (k % MD_TIDM_BATCH) == s.nav[slot].timer;
                md_tidm_decide(nullptr, &s, slot, do_speed_control, front, front_dist, dl.heading[slot]);
            }
        }
        __syncthreads();
        MD_STAMP_AT(2);
        for (int j = tid; j < cap; j += kBlock) {
            if (c.ego_replay && j < A) md_scenario_replay_ego(&s, &c, j, k);   // agent_policy = ReplayEgoCarPolicy
            else if (!(kSkip & 128)) md_integrate_mover(&s, &c, j);
            l_shape_ct[j] = l_shape[j];   // what the contact test below sees
        }
        __syncthreads();
    }
    MD_STAMP_AT(3);
    // The agent's contact flags come from BaseVehicle.after_step (a contact test at the bodies' present poses), which the agent
    // manager runs BEFORE the traffic manager's after_step: replayed bodies are still at frame k-1, bodies removed / spawned in
    // this step are still / not yet there.  Wave 1 tests against a snapshot of the shapes while wave 0 runs after_step and the
    // agent's projection, and waves 2 / 3 the detectors: one stage, one barrier.
    if (just_reset) {
        copy16(l_shape_ct, l_shape, cap * (int)sizeof(MdShape), tid, kBlock);
        __syncthreads();
    }
    if (wave == 1 && !(kSkip & 4)) {
        MdState sc = s;
        sc.shape = l_shape_ct;
        for (int a = 0; a < A; ++a) contacts_vehicle(w, sc, c, e, a, lane, l_cfl);
    }
    // ---- after_step of the traffic manager: lanes = track slots, in slot order ----
    if (wave == 0 && !(kSkip & 64)) {
        for (int j0 = A; j0 < cap; j0 += 64) {
            const int j = j0 + lane;
            const bool in = j < cap;
            const int wants = in ? md_scenario_slot_after_step(&w, &s, &c, e, j, k, 0, 1) : 0;
            const unsigned long long m = __ballot(wants != 0);
            const int before = *l_count + __popcll(m & ((1ull << lane) - 1ull));
            if (in) md_scenario_slot_after_step(&w, &s, &c, e, j, k, before, 0);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) *l_count += __popcll(m);
            wave_sync();
            // a reactive policy created at a frame other than its first run's start: its route is cut at this frame.  Rare
            // (a few per scene and episode): the wave stages the run's remaining positions, lane 0 builds (md_build_route).
            unsigned long long mb = __ballot(in && s.route_n != nullptr && s.route_n[4 * (in ? j : 0) + 3] > 0);
            while (mb) {
                const int jb = j0 + __ffsll((long long)mb) - 1;
                mb &= mb - 1;
                const int kb = s.route_n[4 * jb + 2];
                const int nb = min(s.route_n[4 * jb + 3], c.route_seg_cap);
                for (int i = lane; i < nb; i += 64) {
                    const MdShape fr = s.track_shape[(size_t)(kb + i) * (size_t)c.n_envs * (size_t)cap + (size_t)jb];
                    l_pts[2 * i] = fr.cx;
                    l_pts[2 * i + 1] = fr.cy;
                }
                wave_sync();
                build_route_wave(s.route_n + 4 * jb, s.route_segs + (size_t)jb * c.route_seg_cap, c.route_seg_cap,
                                 s.route_verts + 2 * (size_t)jb * c.route_vert_cap, c.route_vert_cap, s.route_aux + 8 * (size_t)jb, l_pts, nb,
                                 l_link, l_len, lane);
                wave_sync();
            }
        }
    }
    MD_STAMP_AT(4);
    // ---- the agents, in the same stage (nothing below reads a slot after_step writes: the agent's own pose is final after the
    // integration): projection on the reference trajectory (wave 0, after its after_step); waves 2, 3: the detectors ----
    for (int a = 0; a < A; ++a) {
        if (wave == 0) {
            // the reference trajectory = the agent slot's static polyline (md_poly_of): its record is in LDS already
            MdPoly ref;
            ref.segs = uni_p(l_desc[a].segs);
            ref.n = uni_i(l_desc[a].n);
            ref.length = 0.0f;
            MdTrajLoc L;
            float ref_len;
            if (kSkip & 8) {
                L.lng = L.lat = L.heading_at = L.lat_dy = 0.0f;
                L.lat_dx = 1.0f;
                ref_len = 100.0f;
            } else
                traj_locate_wave(ref, s.shape[a].cx, s.shape[a].cy, lane, &L, &ref_len);
            if (lane == 0) {
                l_loc[a] = L;
                l_count[3] = __float_as_int(ref_len);   // single-agent scenes (md_step refuses others): one length
            }
        } else if (wave >= 2 && fused_det && !(kSkip & 2)) {
            // waves 2 and 3 have nothing to do in this stage and the next: the side / lane-line detectors of the agent,
            // each wave one half of the scene's line pieces (the poses are final here)
            const MdShape me = s.shape[a];
            if (md_present(me.flags)) {
                const int mq = w.env_map[e];
                const int q0 = w.quad_off[mq], q1 = w.quad_off[mq + 1];
                const int half = (q1 - q0 + 1) >> 1;
                const int qa = q0 + (wave - 2) * half, qb = min(qa + half, q1);
                int* best = l_dbest + a * n_det;
                const int ns = w.side_beam_cs ? c.n_side : 0;
                int* pairs = reinterpret_cast<int*>(l_beams + 2 * n_det) + (wave - 2) * kDetPairs;   // this wave's pair list
                if (w.side_beam_cs && c.n_side > 0)
                    detector_wave(w, me, qa, qb, l_beams, c.n_side, c.side_range, c.side_mask, best, pairs, lane);
                if (w.ll_beam_cs && c.n_lane_line > 0)
                    detector_wave(w, me, qa, qb, l_beams + 2 * ns, c.n_lane_line, c.ll_range, c.ll_mask, best + ns, pairs, lane);
            }
        }
    }
    __syncthreads();
    MD_STAMP_AT(5);
    // the nine way points of the navigation vector: one lane each (wave 0), both slots of a way point get the value
    if (wave == 0 && lane < MD_TRAJ_NUM_WAY_POINT - 1) {
        const int k0 = w.ckpt_off[e], n_ck = w.ckpt_off[e + 1] - k0;
        const float* ck = w.ckpt_xy + 2 * (size_t)k0;
        for (int a = 0; a < A; ++a) {
            const MdTrajLoc L = l_loc[a];
            const MdShape sh = s.shape[a];
            const float v = md_traj_navi_point(ck, n_ck, md_traj_next_idx(&L, n_ck), lane, sh.cx, sh.cy, sh.c, sh.s);
            float* o = s.obs + (size_t)a * c.obs_dim + md_sc_obs_navi(&c);
            o[2 * lane] = v;
            o[2 * lane + 1] = v;
        }
    }
    if (tid < A) {
        const int a = tid;
        s.flags[a] = l_cfl[a];
        md_scenario_observe_at(&w, &s, &c, e, a, just_reset, &l_loc[a], __int_as_float(l_count[3]), 0);
    }
    MD_STAMP_AT(6);
    if (c.n_beams > 0 && !(kSkip & 1)) phase_lidar(w, s, c, e, tid, kWaves, lidar_out, lidar_stride, lidar_offset, nullptr);
    __syncthreads();
    MD_STAMP_AT(7);
    if (fused_det) {   // after the observation (whose lane 0 filled these dims with "nothing seen")
        const int ns = w.side_beam_cs ? c.n_side : 0;
        for (int it = tid; it < A * n_det; it += kBlock) {
            const int a = it / n_det, i = it - a * n_det;
            float* o = s.obs + (size_t)a * c.obs_dim;
            if (i < ns) o[md_obs_base(&c) + i] = __int_as_float(l_dbest[it]);
            else o[md_obs_ll(&c) + (i - ns)] = __int_as_float(l_dbest[it]);
        }
    }
    copy16(gv.shape, l_shape, cap * (int)sizeof(MdShape), tid, kBlock);
    copy16(gv.dyn, l_dyn, cap * (int)sizeof(MdDyn), tid, kBlock);
    copy16(gv.pid, l_pid, cap * (int)sizeof(MdPid), tid, kBlock);
    copy16(gv.nav, l_nav, cap * (int)sizeof(MdNav), tid, kBlock);
    for (int j = tid; j < cap; j += kBlock) {
        gv.action[2 * j] = l_action[2 * j];
        gv.action[2 * j + 1] = l_action[2 * j + 1];
        gv.flags[j] = l_flags[j];
    }
    if (gv.route_n)
        for (int j = tid; j < 4 * cap; j += kBlock) gv.route_n[j] = l_rn[j];
    if (tid == 0) {
        gv.next_agent_id[0] = *l_count;
        if (do_reset) gv.need_reset[0] = 0;
    }
    MD_STAMP_AT(11);
}

// Lidar.perceive as a sensor on its own (md_lidar_detect): cloud points AND the detected-object sets, from nothing but
// the shape table.  One workgroup per env, shapes + sets in LDS, (agent, sector) items dealt to the waves.
__global__ __launch_bounds__(256) void lidar_detect_kernel(MdWorld w, MdState g, MdConfig c, float* out, int out_stride, int out_offset,
                                                          unsigned long long* detected) {
    const int e = blockIdx.x;
    if (e >= c.n_envs) return;
    const int tid = threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    MdShape* l_shape = reinterpret_cast<MdShape*>(smem);
    unsigned long long* l_det = reinterpret_cast<unsigned long long*>(l_shape + c.cap);
    copy16(l_shape, g.shape + (size_t)e * c.cap, c.cap * (int)sizeof(MdShape), tid, 256);
    for (int j = tid; j < 2 * c.agents_per_env; j += 256) l_det[j] = 0ull;
    __syncthreads();
    MdState s = g;
    s.shape = l_shape;
    phase_lidar(w, s, c, e, tid, 4, out, out_stride, out_offset, l_det);
    __syncthreads();
    for (int j = tid; j < 2 * c.agents_per_env; j += 256) detected[(size_t)e * c.agents_per_env * 2 + j] = l_det[j];
}

// random_traffic with auto-reset (PGTrafficManager with `random_traffic`: the traffic stream is not re-seeded at reset, every episode
// sees other traffic, manager/traffic_manager.py:335-337): the caller stages n_draws host-built traffic draws on the device; an env
// that has just finished its episode (need_reset != 0: md_step restores it from the snapshot at the NEXT step) takes the next draw --
// the snapshot rows and the per-slot constants of the traffic (parameters, routes, pre-drawn lane-change timers) of that env are
// replaced.  One workgroup per env, a no-op for the envs that go on.
__global__ __launch_bounds__(256) void swap_draw_kernel(MdState live, MdState staged, MdConfig c, int n_draws, int32_t* draw_idx) {
    const int e = blockIdx.x;
    if (e >= c.n_envs || live.need_reset[e] == 0) return;   // block-uniform
    const int tid = threadIdx.x;
    const int k = (draw_idx[e] + 1) % n_draws;
    __syncthreads();   // every thread has read the index
    if (tid == 0) draw_idx[e] = k;
    const size_t row = (size_t)e * c.cap, from = ((size_t)k * c.n_envs + e) * c.cap;
    auto words = [&](void* dst, const void* src, size_t elem_words) {   // cap slots of elem_words 4-byte words each
        if (dst == nullptr || src == nullptr) return;
        uint32_t* d = reinterpret_cast<uint32_t*>(dst) + row * elem_words;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(src) + from * elem_words;
        for (size_t i = tid; i < (size_t)c.cap * elem_words; i += 256) d[i] = q[i];
    };
    words(const_cast<MdShape*>(live.shape0), staged.shape0, sizeof(MdShape) / 4);
    words(const_cast<MdDyn*>(live.dyn0), staged.dyn0, sizeof(MdDyn) / 4);
    words(const_cast<MdNav*>(live.nav0), staged.nav0, sizeof(MdNav) / 4);
    words(const_cast<MdPid*>(live.pid0), staged.pid0, sizeof(MdPid) / 4);
    words(live.param, staged.param, sizeof(MdParam) / 4);
    words(const_cast<MdParam*>(live.param0), staged.param, sizeof(MdParam) / 4);
    words(live.route_nodes, staged.route_nodes, MD_ROUTE_LEN);
    words(const_cast<int32_t*>(live.route_nodes0), staged.route_nodes, MD_ROUTE_LEN);
    words(live.route_roads, staged.route_roads, MD_ROUTE_LEN);
    words(const_cast<int32_t*>(live.route_roads0), staged.route_roads, MD_ROUTE_LEN);
    words(live.final_lane, staged.final_lane, 1);
    words(const_cast<int32_t*>(live.final_lane0), staged.final_lane, 1);
    words(const_cast<int32_t*>(live.idm_rand), staged.idm_rand, MD_IDM_RAND);
}

// "Others" block of the observation (Lidar.get_surrounding_vehicles_info): one thread per agent, after the
// step kernel has written the detected sets and the new state back.  Off in the headline configs.
__global__ __launch_bounds__(64) void others_kernel(MdWorld w, MdState g, MdConfig c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n_envs * c.agents_per_env) return;
    const int e = i / c.agents_per_env, a = i - e * c.agents_per_env;
    const MdState s = md_env_view(&g, &c, e);
    const int m = w.env_map[e];
    md_others_block(w.lanes + w.lane_off[m], w.roads + w.road_off[m], &s, &c, a, s.detected[2 * a], s.detected[2 * a + 1],
                    s.obs + (size_t)a * c.obs_dim + md_obs_others(&c));
}

__global__ void probe_kernel(int op, const float* a, const float* b, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = md_probe_eval(op, a[i], b[i]);
}

int check_common(const MdWorld* w, const MdState* s, const MdConfig* c) {
    if (!w || !s || !c) {
        snprintf(g_err, sizeof g_err, "null MdWorld/MdState/MdConfig pointer");
        return MD_EINVAL;
    }
    if (c->struct_size != (int32_t)sizeof(MdConfig)) {
        snprintf(g_err, sizeof g_err, "MdConfig.struct_size=%d, library expects %d", c->struct_size, (int)sizeof(MdConfig));
        return MD_EABI;
    }
    if (c->n_envs <= 0 || c->cap <= 0 || c->cap > MD_MAX_CAP || c->agents_per_env <= 0 || c->agents_per_env > c->cap) {
        snprintf(g_err, sizeof g_err, "bad sizes: n_envs=%d cap=%d agents_per_env=%d (cap<=%d)", c->n_envs, c->cap,
                 c->agents_per_env, MD_MAX_CAP);
        return MD_EINVAL;
    }
    if (c->n_beams < 0 || c->n_beams > MD_MAX_BEAMS) {
        snprintf(g_err, sizeof g_err, "n_beams=%d out of range [0,%d]", c->n_beams, MD_MAX_BEAMS);
        return MD_EINVAL;
    }
    if (!s->shape) {
        snprintf(g_err, sizeof g_err, "MdState.shape is null");
        return MD_EINVAL;
    }
    if (w->n_envs != c->n_envs) {
        snprintf(g_err, sizeof g_err, "MdWorld.n_envs=%d != MdConfig.n_envs=%d", w->n_envs, c->n_envs);
        return MD_EINVAL;
    }
    return MD_OK;
}

int need(const void* p, const char* name) {
    if (p) return MD_OK;
    snprintf(g_err, sizeof g_err, "required pointer %s is null", name);
    return MD_EINVAL;
}

// md_step of the single-agent envs has two kernels: env_kernel (one 4-wave workgroup per env) and wave_step_kernel (one
// wave per env), chosen by MdConfig.step_kernel alone (0 = workgroup, 1 = wave): the host decides -- it knows how many
// distinct maps the batch shares, which is what tips the balance (metadrive_ped_amd/engine.py) -- and the library reads no
// environment variable.
bool use_wave_kernel(const MdConfig* c) { return !c->is_multi_agent && c->step_kernel == 1; }

int launch_wave_step(const MdWorld* w, const MdState* s, const MdConfig* c, float* lidar_out, int stride, int offset,
                     void* stream) {
    const size_t per_env = (size_t)wave_env_lds_bytes(c->cap, c->agents_per_env);
    int per_wg = kWaveEnvs;   // envs per workgroup: as many as fit 64 KB of LDS (capacity-128 accident scenes: 2)
    while (per_wg > 1 && per_wg * per_env > 64 * 1024) per_wg >>= 1;
    const size_t lds = per_wg * per_env;
    if (lds > 64 * 1024) {
        snprintf(g_err, sizeof g_err, "LDS image of one env needs %zu B (cap=%d); limit 65536", lds, c->cap);
        return MD_EINVAL;
    }
    const dim3 grid((c->n_envs + per_wg - 1) / per_wg);
    const hipStream_t st = (hipStream_t)stream;
    const bool general = c->traffic_mode != 0 || c->agent_idm != 0 || s->detected != nullptr;
    if (general) hipLaunchKernelGGL((wave_step_kernel<true>), grid, dim3(64 * per_wg), lds, st, *w, *s, *c, lidar_out, stride, offset);
    else hipLaunchKernelGGL((wave_step_kernel<false>), grid, dim3(64 * per_wg), lds, st, *w, *s, *c, lidar_out, stride, offset);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(g_err, sizeof g_err, "kernel launch failed: %s", hipGetErrorString(err));
        return MD_ELAUNCH;
    }
    return MD_OK;
}

template <int PH>
int launch(const MdWorld* w, const MdState* s, const MdConfig* c, float* lidar_out, int stride, int offset,
           void* stream) {
    if (PH == PH_ALL && use_wave_kernel(c)) return launch_wave_step(w, s, c, lidar_out, stride, offset, stream);
    const bool stage = w->max_lanes <= kStageMaxLanes;
    constexpr bool kCanMultiLds = (PH & (PH_LIFECYCLE | PH_RESET)) != 0;  // same rule as the MULTI kernel variant below
    // Multi-agent md_step: eight waves per env while every workgroup of the batch is resident at once at that size (the MULTI
    // kernel's 92 VGPRs allow 5 waves per SIMD = 640 eight-wave workgroups on the 256 CUs); larger batches keep four, which
    // then fill the chip by themselves.  Measured: 512 tollgate envs x 40 agents, 1024 roundabout envs (profiles/r03_*).
    const bool wide = (PH == PH_ALL) && kCanMultiLds && c->is_multi_agent && c->n_envs <= kWideMaxEnvs && MD_ENV_BLOCK == 256 &&
                      c->agents_per_env > 8;
    const int blk = wide ? 512 : MD_ENV_BLOCK;
    // the lidar-only kernel stages nothing but the shapes: asking for the full image would cost it occupancy
    const size_t lds = (PH == PH_LIDAR) ? (size_t)c->cap * sizeof(MdShape) + 16 :
                       (size_t)c->cap * (sizeof(MdShape) + sizeof(MdDyn) + sizeof(MdNav) + sizeof(MdPid) + 8) +
                       (size_t)((c->cap + 3) & ~3) * 4 +
                       (stage ? (size_t)w->max_lanes * sizeof(MdLane) + (size_t)w->max_roads * sizeof(MdRoad) : 0) +
                       (blk / 64) * (kCanMultiLds && c->is_multi_agent ? kObsScratch : 48) * 4 + (size_t)c->cap * (sizeof(MdParam) + 4) + 16 + (size_t)c->agents_per_env * 16 + 8 +
                       (size_t)c->cap * 8 + 32 + 16;   // + 32: the lifecycle's 8 scratch words sit at the start of the last region; + 16: ticket counters
    if (lds > 64 * 1024 || ((PH != PH_LIDAR) && (w->max_lanes <= 0 || w->max_roads <= 0))) {
        snprintf(g_err, sizeof g_err, "LDS image of one env needs %zu B (cap=%d, max_lanes=%d, max_roads=%d); limit 65536",
                 lds, c->cap, w->max_lanes, w->max_roads);
        return MD_EINVAL;
    }
    const dim3 grid(c->n_envs);
    const hipStream_t st = (hipStream_t)stream;
    constexpr bool kCanRespawn = (PH & (PH_TRAFFIC | PH_RESET | PH_INTEGRATE)) != 0;
    constexpr bool kCanMulti = (PH & (PH_LIFECYCLE | PH_RESET)) != 0;
#define MD_LAUNCH(STAGE, RESP, MUL) \
    hipLaunchKernelGGL((env_kernel<PH, STAGE, RESP, MUL>), grid, dim3(MD_ENV_BLOCK), lds, st, *w, *s, *c, lidar_out, stride, offset)
#define MD_LAUNCH_WIDE(STAGE) \
    hipLaunchKernelGGL((env_kernel<PH, STAGE, false, kCanMulti && PH == PH_ALL, (PH == PH_ALL ? 512 : MD_ENV_BLOCK)>), grid, dim3(512), lds, st, *w, *s, *c, lidar_out, stride, offset)
    if (kCanMulti && c->is_multi_agent && wide) {
        if (stage) MD_LAUNCH_WIDE(true);
        else MD_LAUNCH_WIDE(false);
    } else if (kCanMulti && c->is_multi_agent) {
        if (stage) MD_LAUNCH(true, false, kCanMulti);
        else MD_LAUNCH(false, false, kCanMulti);
    } else if (kCanRespawn && (c->traffic_mode != 0 || c->agent_idm != 0 || (PH == PH_ALL && s->detected != nullptr))) {
        if (stage) MD_LAUNCH(true, kCanRespawn, false);
        else MD_LAUNCH(false, kCanRespawn, false);
    } else if (stage) {
        MD_LAUNCH(true, false, false);
    } else {
        MD_LAUNCH(false, false, false);
    }
#undef MD_LAUNCH
#undef MD_LAUNCH_WIDE
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(g_err, sizeof g_err, "kernel launch failed: %s", hipGetErrorString(err));
        return MD_ELAUNCH;
    }
    return MD_OK;
}

#define NEED(p)                                  \
    do {                                         \
        int _r = need((const void*)(p), #p);     \
        if (_r != MD_OK) return _r;              \
    } while (0)

int check_state(const MdState* s) {
    NEED(s->shape); NEED(s->dyn); NEED(s->param); NEED(s->nav); NEED(s->pid); NEED(s->action); NEED(s->flags);
    NEED(s->route_nodes); NEED(s->route_roads); NEED(s->need_reset);
    return MD_OK;
}

int check_world(const MdWorld* w) {
    NEED(w->env_map); NEED(w->lane_off); NEED(w->lanes); NEED(w->hull_xy); NEED(w->road_off); NEED(w->roads);
    NEED(w->quad_off); NEED(w->quads); NEED(w->quad_kind); NEED(w->grid); NEED(w->cell_start); NEED(w->cell_items);
    return MD_OK;
}

// traffic_mode respawn / hybrid: the respawn-lane tables, the env RNG and the route snapshot must be there
int check_traffic_mode(const MdWorld* w, const MdState* s, const MdConfig* c) {
    if (c->traffic_mode == 0 || c->traffic_mode == 4) return MD_OK;   // 4 (scenario): validated by md_step itself
    if (c->traffic_mode < 0 || c->traffic_mode > 3 || c->is_multi_agent) {
        snprintf(g_err, sizeof g_err, "traffic_mode=%d is not valid here (0 trigger, 1 respawn, 2 hybrid, 3 replay; "
                 "single-agent envs)", c->traffic_mode);
        return MD_EINVAL;
    }
    if (c->traffic_mode == 3) {
        NEED(s->track_shape); NEED(s->track_dyn); NEED(s->nav);
        if (c->track_len <= 0) {
            snprintf(g_err, sizeof g_err, "traffic_mode 3 (replay) needs MdConfig.track_len > 0");
            return MD_EINVAL;
        }
        return MD_OK;
    }
    int r = check_world(w);
    if (r != MD_OK) return r;
    NEED(s->rng); NEED(s->route_nodes0); NEED(s->route_roads0); NEED(s->final_lane0); NEED(s->final_lane);
    NEED(w->spawn_off); NEED(w->spawn_lane); NEED(w->spawn_route); NEED(w->spawn_route_meta);
    return MD_OK;
}


}  // namespace

extern "C" {

__attribute__((visibility("default"))) int md_abi(int32_t* sizes, int n) {
    const int32_t v[11] = {sizeof(MdShape), sizeof(MdDyn), sizeof(MdParam), sizeof(MdNav), sizeof(MdPid), sizeof(MdLane),
                           sizeof(MdRoad), sizeof(MdGrid), sizeof(MdWorld), sizeof(MdState), sizeof(MdConfig)};
    for (int i = 0; sizes && i < n && i < 11; ++i) sizes[i] = v[i];
    return MD_ABI_VERSION;
}

__attribute__((visibility("default"))) const char* md_last_error(void) { return g_err; }

#ifdef MD_STAMP
// diagnostic build only: buffer of n_envs * 16 uint64 device words, or NULL to stop stamping
__attribute__((visibility("default"))) int md_debug_set_stamp_buffer(void* dev_ptr) {
    unsigned long long* p = (unsigned long long*)dev_ptr;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)) == hipSuccess ? MD_OK : MD_ELAUNCH;
}
// diagnostic build only: a permutation of the envs (workgroup b steps env order[b]), or NULL
__attribute__((visibility("default"))) int md_debug_set_env_order(const void* dev_ptr) {
    const int* p = (const int*)dev_ptr;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_env_order), &p, sizeof(p)) == hipSuccess ? MD_OK : MD_ELAUNCH;
}
#endif

// 4 x 16 B per thread in flight (loads issued before the stores), non-temporal both ways, 2048 workgroups
// striding over 16 KB tiles: the usual shape of a bandwidth probe on this part.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n16) {
    constexpr int U = 4;
    const size_t tile = (size_t)256 * U;
    const size_t n_tiles = n16 / tile;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t base = t * tile + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&src[base + (size_t)u * 256]);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], &dst[base + (size_t)u * 256]);
    }
    // tail (< one tile): first workgroup
    if (blockIdx.x == 0)
        for (size_t i = n_tiles * tile + threadIdx.x; i < n16; i += 256) dst[i] = src[i];
}

__attribute__((visibility("default"))) int md_probe_stream_copy(void* dst, const void* src, size_t nbytes, void* stream) {
    if (!dst || !src || nbytes == 0 || (nbytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) {
        snprintf(g_err, sizeof g_err, "md_probe_stream_copy: null / unaligned pointer or size not a multiple of 16");
        return MD_EINVAL;
    }
    const size_t n16 = nbytes >> 4;
    size_t blocks = (n16 + 1023) / 1024;
    if (blocks > 2048u) blocks = 2048u;  // 8 workgroups per CU, tile-stride beyond
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (u32x4*)dst,
                       (const u32x4*)src, n16);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(g_err, sizeof g_err, "kernel launch failed: %s", hipGetErrorString(err));
        return MD_ELAUNCH;
    }
    return MD_OK;
}

__attribute__((visibility("default"))) int md_probe_math(int op, const float* a, const float* b, float* out, int n,
                                                        void* stream) {
    if (!a || !b || !out || n <= 0) {
        snprintf(g_err, sizeof g_err, "md_probe_math: null pointer or n<=0");
        return MD_EINVAL;
    }
    hipLaunchKernelGGL(probe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, op, a, b, out, n);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(g_err, sizeof g_err, "kernel launch failed: %s", hipGetErrorString(err));
        return MD_ELAUNCH;
    }
    return MD_OK;
}

__attribute__((visibility("default"))) int md_lidar(const MdWorld* w, const MdState* s, const MdConfig* c, float* out,
                                                   int out_stride, int out_offset, void* stream) {
    int r = check_common(w, s, c);
    if (r != MD_OK) return r;
    NEED(out); NEED(w->beam_cs);
    if (c->n_beams <= 0 || out_stride < c->n_beams + out_offset || out_offset < 0) {
        snprintf(g_err, sizeof g_err, "md_lidar: n_beams=%d stride=%d offset=%d", c->n_beams, out_stride, out_offset);
        return MD_EINVAL;
    }
    return launch<PH_LIDAR>(w, s, c, out, out_stride, out_offset, stream);
}

__attribute__((visibility("default"))) int md_lidar_detect(const MdWorld* w, const MdState* s, const MdConfig* c, float* out,
                                                          int out_stride, int out_offset, uint64_t* detected, void* stream) {
    int r = check_common(w, s, c);
    if (r != MD_OK) return r;
    NEED(out); NEED(detected); NEED(w->beam_cs);
    if (c->n_beams <= 0 || out_stride < c->n_beams + out_offset || out_offset < 0) {
        snprintf(g_err, sizeof g_err, "md_lidar_detect: n_beams=%d stride=%d offset=%d", c->n_beams, out_stride, out_offset);
        return MD_EINVAL;
    }
    const size_t lds = (size_t)c->cap * sizeof(MdShape) + (size_t)c->agents_per_env * 16 + 16;
    hipLaunchKernelGGL(lidar_detect_kernel, dim3(c->n_envs), dim3(256), lds, (hipStream_t)stream, *w, *s, *c, out, out_stride,
                       out_offset, (unsigned long long*)detected);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(g_err, sizeof g_err, "kernel launch failed: %s", hipGetErrorString(err));
        return MD_ELAUNCH;
    }
    return MD_OK;
}

static int line_detector_launch(const MdWorld* w, const MdState* s, const MdConfig* c, const float* beam_cs, int n_beams, float range,
                                uint32_t kind_mask, int out_offset, const float* beam_cs1, int n_beams1, float range1, uint32_t kind_mask1,
                                int out_offset1, float* out, int out_stride, void* stream, const char* who) {
    int r = check_common(w, s, c);
    if (r != MD_OK) return r;
    NEED(out); NEED(beam_cs); NEED(w->env_map); NEED(w->quad_off); NEED(w->quads); NEED(w->quad_kind);
    if (n_beams <= 0 || n_beams > MD_MAX_BEAMS || out_stride < n_beams + out_offset || out_offset < 0 || !(range > 0.0f) ||
        (n_beams1 > 0 && (!beam_cs1 || out_stride < n_beams1 + out_offset1 || out_offset1 < 0 || !(range1 > 0.0f)))) {
        snprintf(g_err, sizeof g_err, "%s: n_beams=%d/%d stride=%d offset=%d/%d range=%f/%f", who, n_beams, n_beams1, out_stride, out_offset,
                 out_offset1, (double)range, (double)range1);
        return MD_EINVAL;
    }
    const size_t nb = (size_t)n_beams + (size_t)(n_beams1 > 0 ? n_beams1 : 0);
    if (nb > 255) {   // a (quad, beam) pair keeps the beam in eight bits
        snprintf(g_err, sizeof g_err, "%s: %zu beams > 255", who, nb);
        return MD_EINVAL;
    }
    const size_t lds_ld = ((kBlock / 64) * nb + 2 * nb + (kBlock / 64) * (size_t)kDetPairs) * sizeof(int);
    hipLaunchKernelGGL(line_detector_kernel, dim3(c->n_envs * detector_groups(c->agents_per_env)), dim3(kBlock), lds_ld, (hipStream_t)stream, *w, *s, *c, beam_cs,
                       n_beams, range, kind_mask, out, out_stride, out_offset, beam_cs1, n_beams1 > 0 ? n_beams1 : 0, range1, kind_mask1, out_offset1);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(g_err, sizeof g_err, "kernel launch failed: %s", hipGetErrorString(err));
        return MD_ELAUNCH;
    }
    return MD_OK;
}

__attribute__((visibility("default"))) int md_line_detector(const MdWorld* w, const MdState* s, const MdConfig* c,
                                                           const float* beam_cs, int n_beams, float range,
                                                           uint32_t kind_mask, float* out, int out_stride,
                                                           int out_offset, void* stream) {
    return line_detector_launch(w, s, c, beam_cs, n_beams, range, kind_mask, out_offset, nullptr, 0, 0.0f, 0u, 0, out, out_stride, stream,
                                "md_line_detector");
}

// Two detector fans (the side detector and the lane-line detector of one observation) in ONE launch and one pass over the map's
// line pieces: what SideDetector.perceive + LaneLineDetector.perceive cost twice (obs/state_obs.py:77-86,129-140).
__attribute__((visibility("default"))) int md_swap_draw(const MdState* s, const MdState* staged, const MdConfig* c, int n_draws,
                                                        int32_t* draw_idx, void* stream) {
    if (!s || !staged || !c || !draw_idx) {
        snprintf(g_err, sizeof g_err, "md_swap_draw: null MdState / staged MdState / MdConfig / draw_idx");
        return MD_EINVAL;
    }
    if (c->struct_size != (int32_t)sizeof(MdConfig)) {
        snprintf(g_err, sizeof g_err, "MdConfig.struct_size=%d, library expects %d", c->struct_size, (int)sizeof(MdConfig));
        return MD_EABI;
    }
    if (n_draws < 1 || c->n_envs <= 0 || c->cap <= 0 || c->cap > MD_MAX_CAP) {
        snprintf(g_err, sizeof g_err, "md_swap_draw: n_draws=%d n_envs=%d cap=%d", n_draws, c->n_envs, c->cap);
        return MD_EINVAL;
    }
    if (!s->need_reset || !s->shape0 || !s->dyn0 || !s->nav0 || !s->pid0 || !staged->shape0 || !staged->dyn0 || !staged->nav0 || !staged->pid0) {
        snprintf(g_err, sizeof g_err, "md_swap_draw: need_reset and the snapshot arrays shape0 / dyn0 / nav0 / pid0 (live and staged) are required");
        return MD_EINVAL;
    }
    hipLaunchKernelGGL(swap_draw_kernel, dim3(c->n_envs), dim3(256), 0, (hipStream_t)stream, *s, *staged, *c, n_draws, draw_idx);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(g_err, sizeof g_err, "kernel launch failed: %s", hipGetErrorString(err));
        return MD_ELAUNCH;
    }
    return MD_OK;
}

__attribute__((visibility("default"))) int md_line_detectors(const MdWorld* w, const MdState* s, const MdConfig* c,
                                                            const float* beam_cs0, int n_beams0, float range0, uint32_t kind_mask0,
                                                            int out_offset0, const float* beam_cs1, int n_beams1, float range1,
                                                            uint32_t kind_mask1, int out_offset1, float* out, int out_stride,
                                                            void* stream) {
    return line_detector_launch(w, s, c, beam_cs0, n_beams0, range0, kind_mask0, out_offset0, beam_cs1, n_beams1, range1, kind_mask1,
                                out_offset1, out, out_stride, stream, "md_line_detectors");
}

__attribute__((visibility("default"))) int md_integrate(const MdWorld* w, const MdState* s, const MdConfig* c,
                                                       void* stream) {
    int r = check_common(w, s, c);
    if (r != MD_OK) return r;
    r = check_state(s);
    if (r != MD_OK) return r;
    NEED(s->dyn); NEED(s->param); NEED(s->action);
    r = check_traffic_mode(w, s, c);
    if (r != MD_OK) return r;
    return launch<PH_INTEGRATE>(w, s, c, nullptr, 0, 0, stream);
}

__attribute__((visibility("default"))) int md_localize(const MdWorld* w, const MdState* s, const MdConfig* c,
                                                      void* stream) {
    int r = check_common(w, s, c);
    if (r != MD_OK) return r;
    r = check_state(s);
    if (r != MD_OK) return r;
    r = check_world(w);
    if (r != MD_OK) return r;
    NEED(s->nav); NEED(s->flags); NEED(s->route_nodes); NEED(s->route_roads);
    return launch<PH_LOCALIZE>(w, s, c, nullptr, 0, 0, stream);
}

__attribute__((visibility("default"))) int md_contacts(const MdWorld* w, const MdState* s, const MdConfig* c,
                                                      void* stream) {
    int r = check_common(w, s, c);
    if (r != MD_OK) return r;
    r = check_state(s);
    if (r != MD_OK) return r;
    r = check_world(w);
    if (r != MD_OK) return r;
    NEED(s->flags);
    return launch<PH_CONTACTS>(w, s, c, nullptr, 0, 0, stream);
}

__attribute__((visibility("default"))) int md_observe(const MdWorld* w, const MdState* s, const MdConfig* c,
                                                     void* stream) {
    int r = check_common(w, s, c);
    if (r != MD_OK) return r;
    r = check_state(s);
    if (r != MD_OK) return r;
    r = check_world(w);
    if (r != MD_OK) return r;
    NEED(s->dyn); NEED(s->param); NEED(s->nav); NEED(s->pid); NEED(s->action); NEED(s->route_roads); NEED(s->final_lane);
    NEED(s->flags); NEED(s->obs); NEED(s->reward); NEED(s->cost); NEED(s->step_info); NEED(s->need_reset);
    if (c->obs_dim < md_obs_lidar(c)) {
        snprintf(g_err, sizeof g_err, "obs_dim=%d < %d", c->obs_dim, md_obs_lidar(c));
        return MD_EINVAL;
    }
    return launch<PH_OBSERVE>(w, s, c, nullptr, 0, 0, stream);
}

__attribute__((visibility("default"))) int md_idm(const MdWorld* w, const MdState* s, const MdConfig* c, void* stream) {
    int r = check_common(w, s, c);
    if (r != MD_OK) return r;
    r = check_state(s);
    if (r != MD_OK) return r;
    r = check_world(w);
    if (r != MD_OK) return r;
    NEED(s->dyn); NEED(s->nav); NEED(s->pid); NEED(s->action); NEED(s->route_roads); NEED(s->idm_rand);
    NEED(w->node_adj_off); NEED(w->node_adj); NEED(w->node_off);
    return launch<PH_IDM>(w, s, c, nullptr, 0, 0, stream);
}

__attribute__((visibility("default"))) int md_traffic_after_step(const MdWorld* w, const MdState* s, const MdConfig* c,
                                                                void* stream) {
    int r = check_common(w, s, c);
    if (r != MD_OK) return r;
    r = check_state(s);
    if (r != MD_OK) return r;
    NEED(s->flags);
    r = check_traffic_mode(w, s, c);
    if (r != MD_OK) return r;
    return launch<PH_TRAFFIC>(w, s, c, nullptr, 0, 0, stream);
}

int check_marl(const MdWorld* w, const MdState* s, const MdConfig* c) {
    if (!c->is_multi_agent) return MD_OK;
    NEED(s->rng); NEED(s->env_steps); NEED(s->agent_id); NEED(s->next_agent_id); NEED(s->route_nodes0);
    NEED(s->route_roads0); NEED(s->final_lane0); NEED(s->final_lane);
    if (c->allow_respawn) {
        NEED(w->spawn_off); NEED(w->spawn_place); NEED(w->spawn_lane); NEED(w->spawn_route); NEED(w->spawn_route_meta);
        if (w->n_dest <= 0) {
            snprintf(g_err, sizeof g_err, "multi-agent respawn needs MdWorld.n_dest > 0");
            return MD_EINVAL;
        }
    }
    return MD_OK;
}

__attribute__((visibility("default"))) int md_lifecycle(const MdWorld* w, const MdState* s, const MdConfig* c, void* stream) {
    int r = check_common(w, s, c);
    if (r != MD_OK) return r;
    r = check_state(s);
    if (r != MD_OK) return r;
    r = check_world(w);
    if (r != MD_OK) return r;
    r = check_marl(w, s, c);
    if (r != MD_OK) return r;
    return launch<PH_LIFECYCLE>(w, s, c, nullptr, 0, 0, stream);
}

__attribute__((visibility("default"))) int md_step(const MdWorld* w, const MdState* s, const MdConfig* c, void* stream) {
    int r = check_common(w, s, c);
    if (r != MD_OK) return r;
    r = check_state(s);
    if (r != MD_OK) return r;
    r = check_world(w);
    if (r != MD_OK) return r;
    NEED(s->dyn); NEED(s->param); NEED(s->nav); NEED(s->pid); NEED(s->action); NEED(s->route_nodes); NEED(s->route_roads);
    NEED(s->final_lane); NEED(s->idm_rand); NEED(s->flags); NEED(s->obs); NEED(s->reward); NEED(s->cost);
    NEED(s->step_info); NEED(s->need_reset); NEED(w->node_adj_off); NEED(w->node_adj); NEED(w->node_off);
    if (c->n_beams > 0) NEED(w->beam_cs);
    NEED(s->shape0); NEED(s->dyn0); NEED(s->nav0); NEED(s->pid0);
    r = check_marl(w, s, c);
    if (r != MD_OK) return r;
    r = check_traffic_mode(w, s, c);
    if (r != MD_OK) return r;
    if (c->traffic_mode == 4) {   // scenario mode: its own kernel (ScenarioEnv step)
        NEED(w->poly_off); NEED(w->segs); NEED(w->polyv_off); NEED(w->polyv); NEED(w->ckpt_off); NEED(w->ckpt_xy);
        NEED(w->track_meta); NEED(s->track_shape); NEED(s->track_dyn); NEED(s->next_agent_id);
        if (c->is_multi_agent || c->agents_per_env != 1 || c->track_len <= 0) {
            snprintf(g_err, sizeof g_err, "scenario mode: single-agent scenes with track_len > 0 (got agents=%d track_len=%d)",
                     c->agents_per_env, c->track_len);
            return MD_EINVAL;
        }
        if (c->obs_dim != md_sc_obs_lidar(c) + c->n_beams) {
            snprintf(g_err, sizeof g_err, "scenario mode: obs_dim=%d != %d state/navi dims + n_beams=%d", c->obs_dim,
                     md_sc_obs_lidar(c), c->n_beams);
            return MD_EINVAL;
        }
        const size_t lds = (size_t)c->cap * (4 * 32 + 64 + 8 + 4) + (size_t)((c->cap + 3) & ~3) * 4 +
                           (size_t)c->agents_per_env * sizeof(MdTrajLoc) + 16 +
                           (size_t)(c->agents_per_env + 2) * (size_t)(c->n_side + c->n_lane_line) * sizeof(int) +
                           ((c->n_side + c->n_lane_line) > 0 ? 2 * (size_t)kDetPairs * sizeof(int) : 0) +
                           sc_scratch_bytes(s->route_n != nullptr, c->route_seg_cap, c->cap) + 16 +   // the decision stage's scratch / positions, links, lengths of a route being built
                           (s->route_n ? (size_t)c->cap * 16 : 0) +   // route_n
                           (size_t)c->cap * (sizeof(MdShape) + sizeof(RouteDesc)) + 32;   // the shapes the contact test sees; the route records
        if (s->route_n) {
            NEED(s->route_segs); NEED(s->route_verts); NEED(s->route_aux); NEED(w->run_off); NEED(w->runs);
            if (c->route_seg_cap < 1 || c->route_vert_cap < 8) {
                snprintf(g_err, sizeof g_err, "scenario mode: route buffers need route_seg_cap >= 1 and route_vert_cap >= 8 (got %d, %d)",
                         c->route_seg_cap, c->route_vert_cap);
                return MD_EINVAL;
            }
        }
        if (lds > 64 * 1024) {
            snprintf(g_err, sizeof g_err, "scenario mode: LDS image needs %zu B (cap=%d)", lds, c->cap);
            return MD_EINVAL;
        }
        hipLaunchKernelGGL(scenario_step_kernel, dim3(c->n_envs), dim3(256), lds, (hipStream_t)stream, *w, *s, *c, s->obs, c->obs_dim,
                           md_sc_obs_lidar(c));
        hipError_t err4 = hipGetLastError();
        if (err4 != hipSuccess) {
            snprintf(g_err, sizeof g_err, "kernel launch failed: %s", hipGetErrorString(err4));
            return MD_ELAUNCH;
        }
        return MD_OK;
    }
    if (c->obs_dim != md_obs_lidar(c) + c->n_beams + md_obs_tail(c)) {
        snprintf(g_err, sizeof g_err, "obs_dim=%d != %d state/navi dims + n_beams=%d + %d", c->obs_dim, md_obs_lidar(c), c->n_beams,
                 md_obs_tail(c));
        return MD_EINVAL;
    }
    if (!c->is_multi_agent && c->agents_per_env != 1) {   // the single-agent kernels schedule ONE agent's observation per env
        snprintf(g_err, sizeof g_err, "single-agent envs step one agent per env (agents_per_env=%d): use the multi-agent configs", c->agents_per_env);
        return MD_EINVAL;
    }
    if (c->num_others < 0 || c->num_others > 16 || (c->num_others > 0 && c->n_beams <= 0)) {
        snprintf(g_err, sizeof g_err, "num_others=%d needs 0..16 and the lidar on", c->num_others);
        return MD_EINVAL;
    }
    if (c->num_others > 0) NEED(s->detected);
    r = launch<PH_ALL>(w, s, c, s->obs, c->obs_dim, md_obs_lidar(c), stream);
    if (r != MD_OK || c->num_others <= 0) return r;
    const int n = c->n_envs * c->agents_per_env;
    hipLaunchKernelGGL(others_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, *w, *s, *c);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(g_err, sizeof g_err, "kernel launch failed: %s", hipGetErrorString(err));
        return MD_ELAUNCH;
    }
    return MD_OK;
}

}  // extern "C"
