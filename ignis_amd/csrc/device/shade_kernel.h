// shade_kernel.h — the k_shade kernel template (sort by material + shade + compact, see shade.hip), shared by the translation units
// that instantiate it: shade.hip (path tracer variants, debug views, light tracer) and photon.hip (the photon mapper's two passes).
#pragma once

#include "shade_core.h"
#include "lt_core.h"
#include "ppm_core.h"

namespace igdev {

// ---------------------------------------------------------------- k_shade

#ifndef IG_SHADE_PREFETCH
#define IG_SHADE_PREFETCH 1
#endif
#ifndef IG_SHADE_THREADS
#define IG_SHADE_THREADS 256 // (experiments: the expression kernels' LDS register file is sized for 256)
#endif
constexpr int kShadeThreads = IG_SHADE_THREADS;
constexpr int kMaxSortBins  = 254; // material_count + 2 bins must fit one entry per thread

// Waves per SIMD the full variant is built for. Its natural register demand is 252 VGPRs (215 without the principled BSDF, 196
// without blends, 174 with the lean BSDFs but every light model): 4 waves (128 VGPRs) spill 532 B, 3 waves (168) 300 B, 2 waves
// (256) nothing. Measured on diamond_scene_principled (32 steps): 149 / 135 / 139 ms of shading at 4 / 3 / 2 waves.
#ifndef IG_SHADE_OCC_FULL
#define IG_SHADE_OCC_FULL 3
#endif
#ifndef IG_SHADE_OCC_LEAN
#define IG_SHADE_OCC_LEAN 4
#endif
#ifndef IG_SHADE_OCC_BASIC
// the by-class kernel of the basic models + misses waits for memory (light-hierarchy descents, environment lookups: waves waiting 64 % of their
// cycles on many_point_lights, profiles/r06_traffic_many_point_lights.json): four waves per SIMD at 128 VGPRs. many_point_lights + 2.5 %, the
// principled scene + 1.1 %, the divergent stand-in + 0 (profiles/r06_experiment_ab.txt section 7; round 5 had measured + 1 % / - 0.5 % and kept 3)
#define IG_SHADE_OCC_BASIC 4
#endif
constexpr int kBounceBins = 16; // the bounce rays of a window leave grouped by (specular bounce, octant of the direction)

// The full path-tracer variant by material class: one instantiation per group of BSDF models, each
// launched over its own run of the round's hits, which k_bin_count / k_bin_scan / k_bin_scatter (shade.hip) sort by material globally —
// class-major, so a class is one contiguous run of the index list — as the reference's gpu_sort_primary does (mapping_gpu.art:409-502).
// What a kernel's register file has to hold is then the largest model of its group, not of the whole library: the one-for-all
// instantiation needs 252 VGPRs (168 + 292 B of scratch at three waves per SIMD). Misses go with the basic group. A blend carries
// every model. (Round 4 launched every class over every hit and let a window-local sort park the other classes' rays: each window
// re-sorted four times and most of its lanes idled)
constexpr uint32_t kClassMiss       = 1u << 31;
constexpr uint32_t kClassBasic      = (1u << IG_BSDF_DIFFUSE) | (1u << IG_BSDF_DIELECTRIC) | (1u << IG_BSDF_CONDUCTOR) | (1u << IG_BSDF_TRANSPARENT) | (1u << IG_BSDF_PHONG) | kClassMiss;
constexpr uint32_t kClassPrincipled = 1u << IG_BSDF_PRINCIPLED;
constexpr uint32_t kClassCoated     = (1u << IG_BSDF_PLASTIC) | (1u << IG_BSDF_ROUGH_DIELECTRIC);
constexpr uint32_t kClassBlend      = 1u << IG_BSDF_BLEND;
constexpr uint32_t kClassAll        = ~0u;

// LT: the light tracer's callbacks (lt_core.h) instead of the path tracer's; PPM: the photon mapper's light (1) or camera (2) pass (ppm_core.h)
template <bool FULL, bool DEBUG_VIEWS = false, bool EXPR = false, bool LT = false, int PPM = 0, uint32_t TYPES = kClassAll>
__global__ void __launch_bounds__(kShadeThreads, FULL ? (EXPR ? 2 : (TYPES == kClassBasic ? IG_SHADE_OCC_BASIC : IG_SHADE_OCC_FULL)) : IG_SHADE_OCC_LEAN) k_shade(const ShadeArgs a)
{
    constexpr bool BY_CLASS = TYPES != kClassAll;
    __shared__ uint32_t s_hist[kShadeThreads];
    __shared__ uint32_t s_scan[kShadeThreads];
    __shared__ uint16_t s_perm[kShadeThreads];
    __shared__ uint32_t s_wave_cnt[2][kShadeThreads / 64];
    __shared__ uint32_t s_base[2];
    __shared__ uint32_t s_bin[kBounceBins], s_binoff[kBounceBins]; // bounce rays of a window are written grouped by (specular bounce, direction octant)

    const int tid  = threadIdx.x;
    const int lane = tid & 63;

    const DevScene& sc = a.scene;
    // BY_CLASS: this launch's rays are a run of the round's hits sorted by material (k_bin_*, shade.hip): entries
    // [cls_range[0], cls_range[0] + cls_range[1]) of sort_idx name them
    // (BY_CLASS without an index list: a scene whose materials all belong to this class — the launch takes the stream as it lies, no sort)
    const bool by_index      = BY_CLASS && a.sort_idx != nullptr;
    const uint32_t n         = by_index ? a.cls_range[1] : *a.in_count;
    const uint32_t cls_first = by_index ? a.cls_range[0] : 0u;
    const int M              = (int)sc.material_count;
    // The lean variant has three BSDF models and waits for memory, not for issue slots (a TEA with one round instead of four changes its
    // time by 1 %, profiles/r03_experiment_shade.txt): the sort's two dependent loads and five barriers in front of every window cost it
    // more (4 %) than the divergence they remove. The full variants sort.
    // The by-class variants get their rays sorted already: dense waves of one material, no window sort.
    const bool do_sort = FULL && !BY_CLASS && (M + 2) <= kMaxSortBins;

    const ShadeFrame fr = a.frame;

#ifdef IG_SHADE_CLOCKS
    PhaseClock clk;
    clk.start();
#else
    NoClock clk;
#endif
    const uint32_t chunks = (n + kShadeThreads - 1) / kShadeThreads;
    // The lean variant waits for memory: its window starts with two dependent round trips (the hit, whose entity decides whether the ray's
    // other columns are read at all, then those columns). The hit row of the NEXT window is fetched while this one is shaded.
    constexpr bool kPrefetchHit = IG_SHADE_PREFETCH && !FULL && !LT && PPM == 0 && !DEBUG_VIEWS;
    // (The ray's other columns the same way — 16 more registers — lose more to the spills at 4 waves per SIMD, or to 3 waves per SIMD, than the
    // round trip is worth: A/B section 27.)
    // (The by-class variants, which start with three — the sorted index, the hit, the columns —, gain nothing from index and hit fetched ahead:
    // they wait for issue slots. A/B section 27.)
    float4 hit_pre = make_float4(0, 0, 0, 0);
    if (kPrefetchHit && blockIdx.x < chunks && blockIdx.x * kShadeThreads + tid < n)
        hit_pre = a.in.hit[blockIdx.x * kShadeThreads + tid];
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const uint32_t base = chunk * kShadeThreads;
        clk.mark(10); // loop overhead (and, for the first window, the kernel's prologue)
        const float4 hit_now = hit_pre;
        if (kPrefetchHit) {
            const uint32_t nj = (chunk + gridDim.x) * kShadeThreads + tid;
            if (chunk + gridDim.x < chunks && nj < n)
                hit_pre = a.in.hit[nj];
        }

        // ---- workgroup-local counting sort by material (miss = bin M, out of range = bin M + 1)
        uint32_t j = base + tid;
        // cleared here, in front of the sort's barriers (or the explicit one below when there is no sort): every wave's
        // atomicAdd on s_bin is then ordered behind the clear, and the previous window's last barrier behind its reads
        if (tid < kBounceBins)
            s_bin[tid] = 0;
        if (do_sort) {
            const uint32_t i = base + tid;
            int key          = M + 1;
            if (i < n) {
                int ent = (int)igm_bits(a.in.hit[i].x), prim_;
                if (a.hit_pack)
                    unpack_hit_ids(a.hit_pack, igm_bits(a.in.hit[i].x), ent, prim_);
                key = ent < 0 ? M : sc.entity_material[ent];
            }
            s_hist[tid] = 0;
            __syncthreads();
            const uint32_t r = atomicAdd(&s_hist[key], 1u);
            __syncthreads();
            // inclusive scan over the 256 bins by the first wave: 4 bins per lane + wave shuffle scan
            if (tid < 64) {
                const uint32_t h0 = s_hist[4 * tid], h1 = s_hist[4 * tid + 1], h2 = s_hist[4 * tid + 2], h3 = s_hist[4 * tid + 3];
                const uint32_t local = h0 + h1 + h2 + h3;
                uint32_t incl        = local;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t up = __shfl_up(incl, off);
                    if (lane >= off)
                        incl += up;
                }
                const uint32_t excl = incl - local;
                s_scan[4 * tid]     = excl + h0;
                s_scan[4 * tid + 1] = excl + h0 + h1;
                s_scan[4 * tid + 2] = excl + h0 + h1 + h2;
                s_scan[4 * tid + 3] = incl;
            }
            __syncthreads();
            const uint32_t start = key ? s_scan[key - 1] : 0u;
            s_perm[start + r]    = (uint16_t)tid;
            __syncthreads();
            j = base + s_perm[tid];
            __syncthreads();
        }
        bool valid = j < n;
        if (by_index && valid)
            j = a.sort_idx[cls_first + j];
        if (!by_index && !LT && PPM == 0 && !DEBUG_VIEWS && a.skip_misses) {
            // a scene without environment lights: a miss adds nothing and ends its path (shade_vertex's on_miss sums over no light), so
            // its columns are not even read — a look at the hit first, a wave whose rays all missed (camera rays past the geometry) goes on
            if (valid) {
                const uint32_t hx = igm_bits(kPrefetchHit ? hit_now.x : a.in.hit[j].x);
                valid             = a.hit_pack ? hx != 0xFFFFFFFFu : (int)hx >= 0;
            }
        }

        clk.mark(0); // the sort
        PathVertexOut out;
        out.bounce = out.shadow = out.has_radiance = false;
        out.b_seed = 0u;
        int ray_id = 0;
        int in_ent_for_bin = 0;
        int s_slot = 0; // light tracer: the accumulator slot of the pixel a connection lands in
        if (valid) {
            PathVertexIn in;
            const bool compact = a.in_kind == kStreamCamera && a.cam_stream.compact; // (kernels.h CameraStream: only rayB is stored)
            const float4 ra = compact ? a.cam_stream.rayA : a.in.rayA[j];
            const float4 rb = a.in.rayB[j], hit = kPrefetchHit ? hit_now : a.in.hit[j];
            const int4 meta = compact ? make_int4((int32_t)(a.cam_stream.first_id + j), (int32_t)IG_RAY_FLAG_CAMERA, (int32_t)a.cam_stream.rnd_counter, 1) : a.in.meta[j];
            // (what the stream's writer left constant is not read: kernels.h kStream*)
            const float4 pay = a.in_kind == kStreamCamera ? make_float4(0, 1, 1, 1) : a.in.pay[j];
            in.ray_id  = ray_id = meta.x;
            in.org     = f3{ ra.x, ra.y, ra.z };
            in.dir     = f3{ rb.x, rb.y, rb.z };
            in.rnd     = (uint32_t)meta.z;
            in.inv_pdf = pay.x;
            in.contrib = Col{ pay.y, pay.z, pay.w };
            in.depth   = meta.w;
            in.eta     = a.in_kind == kStreamShaded ? igm_float((uint32_t)meta.y) : 1.0f;
            in.seed    = a.in_kind == kStreamShaded ? igm_bits(rb.w) : 0u;
            in.ent     = in_ent_for_bin = (int)igm_bits(hit.x);
            in.prim    = (int)igm_bits(hit.y);
            in.t = hit.z, in.u = hit.w;
            if (a.hit_pack) {
                unpack_hit_ids(a.hit_pack, igm_bits(hit.x), in.ent, in.prim);
                in_ent_for_bin = in.ent;
                in.t = hit.y, in.u = hit.z, in.v = hit.w;
            } else {
                in.v = a.in.hit_v[j];
            }
            clk.mark(1); // the ray's columns
            if constexpr (PPM != 0)
                shade_vertex_ppm<PPM == 1>(sc, fr, a.ppm, in, out);
            else if constexpr (LT)
                shade_vertex_lt(sc, fr, LtCamera(a.lt_cam), in, out, s_slot);
            else
                shade_vertex<FULL, DEBUG_VIEWS, EXPR, TYPES>(sc, fr, in, out, clk);
            clk.mark(6); // on_bounce (a miss: everything)
            if (out.has_radiance) {
                // per-sample accumulator; the slot is owned by this ray: a plain read-modify-write (no-return float atomics give the same
                // bits and are slower, profiles/r03_experiment_shade.txt)
                float4* acc = a.accum + ((int64_t)ray_id - a.id_base);
                float4 v    = *acc;
                v.x += out.radiance.r * a.inv_spi;
                v.y += out.radiance.g * a.inv_spi;
                v.z += out.radiance.b * a.inv_spi;
                *acc = v;
                if (a.accum_direct && in.ent >= 0) { // aov_di.splat in on_hit (technique/pathtracer.art:133); on_miss has none
                    float4* di = a.accum_direct + ((int64_t)ray_id - a.id_base);
                    float4 w   = *di;
                    w.x += out.radiance.r * a.inv_spi;
                    w.y += out.radiance.g * a.inv_spi;
                    w.z += out.radiance.b * a.inv_spi;
                    *di = w;
                }
            }
        }

        clk.mark(7); // the accumulator
        // ---- append survivors / shadow rays: ONE atomic per workgroup and queue (replaces K9). A single
        // counter word sustains only ~88 atomics/us, so per-wave appends would serialise the kernel.
        // (Ranks from ballots instead of the LDS atomics, the counts scanned by one wave over a DPP row and the bookkeeping words used
        // alternately, i.e. two barriers per window instead of four, measured no faster on the lean variant — 73.7 against 73.1 ms of
        // shading per 20 steps — and 6 % slower on the full one, whose 64-bit lane masks became 22 more spilled registers:
        // profiles/r03_experiment_shade.txt. What a window waits for here is the reservation's round trip.)
        {
            // Continuation rays leave grouped by the octant of their direction: rays of one octant visit BVH children
            // in the same order, so the next round's traversal waves diverge less. (Order inside the stream is free:
            // nothing downstream depends on it.)
            const unsigned long long ms = __ballot(out.shadow);
            const int wave              = tid >> 6;
            if (lane == 0)
                s_wave_cnt[1][wave] = (uint32_t)__popcll(ms);
            int bkey       = 0;
            uint32_t brank = 0;
            if (!do_sort)
                __syncthreads(); // s_bin was cleared at the top of the window; without the sort there is no barrier in between
            if (out.bounce) {
                // rays that continue through a specular (dielectric) vertex start inside or on a refractive object and
                // walk its BVH first; the others cross the room: two populations with different traversal shapes
                bkey = (out.b_dir.x < 0 ? 1 : 0) | (out.b_dir.y < 0 ? 2 : 0) | (out.b_dir.z < 0 ? 4 : 0) | (out.b_inv_pdf == 0 ? 8 : 0);
                brank = atomicAdd(&s_bin[bkey], 1u);
            }
            __syncthreads();
            clk.mark(8); // ballots, bins, the barrier in front of the reservation
            if (tid == 0) {
                // both queues' sizes live in one 64-bit word (QueueState::Counts): one reservation per window
                uint32_t tb = 0;
                for (int k = 0; k < kBounceBins; ++k) {
                    s_binoff[k] = tb;
                    tb += s_bin[k];
                }
                uint32_t ts = 0;
                for (int w = 0; w < kShadeThreads / 64; ++w)
                    ts += s_wave_cnt[1][w];
                unsigned long long old = 0;
                if (tb | ts)
                    old = atomicAdd(reinterpret_cast<unsigned long long*>(a.out_count), (unsigned long long)tb | ((unsigned long long)ts << 32));
                s_base[0] = (uint32_t)old;
                s_base[1] = (uint32_t)(old >> 32);
            }
            __syncthreads();
            clk.mark(11); // the reservation: one thread's scan of the bins and its atomic with return, the barrier behind it
            uint32_t os = s_base[1];
            for (int w = 0; w < wave; ++w)
                os += s_wave_cnt[1][w];
            if (out.bounce) {
                const uint32_t o = s_base[0] + s_binoff[bkey] + brank;
                a.out.rayA[o] = make_float4(out.b_org.x, out.b_org.y, out.b_org.z, FULL ? out.b_tmin : kRayOffset);
                a.out.rayB[o] = make_float4(out.b_dir.x, out.b_dir.y, out.b_dir.z, igm_float(out.b_seed)); // (kStreamShaded: tmax is FLT_MAX for every bounce ray)
                a.out.meta[o] = make_int4(ray_id, (int32_t)igm_bits(out.b_eta), (int32_t)out.b_rnd, out.b_depth); // (kStreamShaded: eta where the flags were)
                a.out.pay[o]  = make_float4(out.b_inv_pdf, out.b_contrib.r, out.b_contrib.g, out.b_contrib.b);
            }
            if (out.shadow) {
                const uint32_t o = os + (uint32_t)__popcll(ms & ((1ull << lane) - 1ull));
                a.sec.rayA[o] = make_float4(out.s_org.x, out.s_org.y, out.s_org.z, kRayOffset);
                a.sec.rayB[o] = make_float4(out.s_dir.x, out.s_dir.y, out.s_dir.z, out.s_tmax);
                a.sec.col[o]  = make_float4(out.s_col.r, out.s_col.g, out.s_col.b, igm_float((uint32_t)(LT ? s_slot : ray_id)));
                if constexpr (LT)
                    a.sec.path_id[o] = (uint32_t)ray_id;
            }
            __syncthreads(); // s_wave_cnt / s_base are reused by the next chunk
            clk.mark(9); // the stores (drained), the last barrier
        }
    }
#ifdef IG_SHADE_CLOCKS
    if (lane == 0) {
        for (int k = 0; k < 6; ++k) {
            atomicAdd(&a.qs->section_passes[k], clk.acc[k]);
            atomicAdd(&a.qs->section_lanes[k], clk.acc[6 + k]);
        }
    }
#endif
}


} // namespace igdev
