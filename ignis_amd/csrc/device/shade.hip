// shade.hip — ray generation, hit/miss shading + path-tracer callbacks, framebuffer resolve.
//
// Replaces the reference stages K1 (gpu_generate_rays, src/artic/driver/mapping_gpu.art:616-669),
// K3a-e (gpu_sort_primary, :409-502), K4 (gpu_hit_shade, :123-214), K5 (gpu_miss_shade, :237-274)
// and K9 (gpu_compact_primary, :686-711) with three kernels:
//   k_generate : one thread per camera sample, writes the SoA primary stream;
//   k_shade    : workgroup-local counting sort of 256 consecutive hits by material (LDS histogram +
//                scan) so each wave shades one BSDF type, then surface reconstruction, emission/MIS,
//                NEE shadow-ray emission, BSDF sampling + Russian roulette; survivors are appended
//                to the OTHER primary stream with one wave-aggregated atomic (sort, M per-material
//                launches, host scan and compaction of the reference collapse into this kernel);
//   k_resolve  : fixed-order sum of the per-sample accumulators into the framebuffer, so a given
//                seed reproduces the image bit for bit (the reference GPU path uses float atomics).
// Shading arithmetic follows src/artic/{core,bsdf,light,technique,camera} expression by expression
// (citations inline); transcendental functions come from include/ig_detmath.h.
#include "shade_core.h"

namespace igdev {

// ---------------------------------------------------------------- k_generate

__global__ void __launch_bounds__(256) k_generate(const GenerateArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        *a.out_count = a.n;
        atomicAdd(&a.qs->camera_rays, (unsigned long long)a.n);
    }
    if (i >= a.n)
        return;

    // gpu_generate_rays id convention: ray id = pixel * spi + sample, linear over the film
    const int64_t lid     = a.first_local_id + i;
    const int sample      = (int)(lid % a.spi);
    const int64_t lpixel  = lid / a.spi;
    const int x           = (int)(lpixel % a.width);
    const int y           = a.row_offset + (int)(lpixel / a.width) * a.row_stride;

    Tea rnd{ make_seed(sample, a.iteration, a.frame, x, y, a.seed), 1 };
    f3 org, dir;
    float tmin, tmax;
    uint32_t flags;
    if (a.list_rays) {
        // make_list_emitter (driver/emitter.art:18-30): film is #rays x 1, flags 0
        const float* r = a.list_rays + (size_t)x * 8;
        org   = f3{ r[0], r[1], r[2] };
        dir   = f3{ r[3], r[4], r[5] };
        tmin  = r[6];
        tmax  = r[7];
        flags = 0;
    } else {
        // make_camera_emitter (driver/emitter.art:6-16), uniform pixel sampler (sampler/pixel_sampler.art:4-10),
        // make_pixelcoord_from_xy (driver/camera.art:21-29), make_perspective_camera (camera/perspective.art:29-42)
        const float rx = rnd.f32();
        const float ry = rnd.f32();
        const float nx = 2 * ((float)x + rx) / ((float)a.width) - 1;
        const float ny = 1 - 2 * ((float)y + ry) / ((float)a.height);
        const f3 cdir  = f3{ a.cam.dir[0], a.cam.dir[1], a.cam.dir[2] };
        const f3 cup   = f3{ a.cam.up[0], a.cam.up[1], a.cam.up[2] };
        m33 view;
        view.c0 = normalize3(cross3(cdir, cup));
        view.c1 = cup;
        view.c2 = cdir;
        org     = f3{ a.cam.eye[0], a.cam.eye[1], a.cam.eye[2] };
        dir     = normalize3(mul33(view, f3{ a.sx * nx, a.sy * ny, 1 }));
        tmin    = a.cam.near_clip;
        tmax    = a.cam.far_clip;
        flags   = IG_RAY_FLAG_CAMERA;
    }

    a.out.id[i] = (int32_t)lid;
    a.out.ox[i] = org.x, a.out.oy[i] = org.y, a.out.oz[i] = org.z;
    a.out.dx[i] = dir.x, a.out.dy[i] = dir.y, a.out.dz[i] = dir.z;
    a.out.tmin[i]  = tmin;
    a.out.tmax[i]  = tmax;
    a.out.flags[i] = flags;
    a.out.rnd[i]   = rnd.counter;
    // init_pt_raypayload (technique/pathtracer.art:33-38)
    a.out.payload[0][i] = 0;
    a.out.payload[1][i] = 1;
    a.out.payload[2][i] = 1;
    a.out.payload[3][i] = 1;
    a.out.payload[4][i] = 1;
    a.out.payload[5][i] = 1;
}

// ---------------------------------------------------------------- k_shade

constexpr int kShadeThreads = 256;
constexpr int kMaxSortBins  = 254; // material_count + 2 bins must fit one entry per thread

__global__ void __launch_bounds__(kShadeThreads) k_shade(const ShadeArgs a)
{
    __shared__ uint32_t s_hist[kShadeThreads];
    __shared__ uint32_t s_scan[kShadeThreads];
    __shared__ uint16_t s_perm[kShadeThreads];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;

    const DevScene& sc = a.scene;
    const uint32_t n   = *a.in_count;
    const int M        = (int)sc.material_count;
    const bool do_sort = (M + 2) <= kMaxSortBins;

    ShadeFrame fr;
    fr.width = a.width, fr.spi = a.spi;
    fr.iteration = a.iteration, fr.frame = a.frame, fr.seed = a.seed;
    fr.row_offset = a.row_offset, fr.row_stride = a.row_stride;

    const uint32_t chunks = (n + kShadeThreads - 1) / kShadeThreads;
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const uint32_t base = chunk * kShadeThreads;

        // ---- workgroup-local counting sort by material (miss = bin M, out of range = bin M + 1)
        uint32_t j = base + tid;
        if (do_sort) {
            const uint32_t i = base + tid;
            int key          = M + 1;
            if (i < n) {
                const int ent = a.in.ent_id[i];
                key           = ent < 0 ? M : sc.entity_material[ent];
            }
            s_hist[tid] = 0;
            __syncthreads();
            const uint32_t r = atomicAdd(&s_hist[key], 1u);
            __syncthreads();
            // inclusive Hillis-Steele scan over the bins
            uint32_t val = s_hist[tid];
            s_scan[tid]  = val;
            __syncthreads();
#pragma unroll
            for (int off = 1; off < kShadeThreads; off <<= 1) {
                const uint32_t add = tid >= off ? s_scan[tid - off] : 0u;
                __syncthreads();
                val += add;
                s_scan[tid] = val;
                __syncthreads();
            }
            const uint32_t start = key ? s_scan[key - 1] : 0u;
            s_perm[start + r]    = (uint16_t)tid;
            __syncthreads();
            j = base + s_perm[tid];
            __syncthreads();
        }

        PathVertexOut out;
        out.bounce = out.shadow = out.has_radiance = false;
        int ray_id = 0;
        if (j < n) {
            PathVertexIn in;
            in.ray_id  = ray_id = a.in.id[j];
            in.org     = f3{ a.in.ox[j], a.in.oy[j], a.in.oz[j] };
            in.dir     = f3{ a.in.dx[j], a.in.dy[j], a.in.dz[j] };
            in.rnd     = a.in.rnd[j];
            in.inv_pdf = a.in.payload[0][j];
            in.contrib = Col{ a.in.payload[1][j], a.in.payload[2][j], a.in.payload[3][j] };
            in.depth   = (int)a.in.payload[4][j];
            in.eta     = a.in.payload[5][j];
            in.ent     = a.in.ent_id[j];
            in.prim    = a.in.prim_id[j];
            in.t = a.in.t[j], in.u = a.in.u[j], in.v = a.in.v[j];
            shade_vertex(sc, fr, in, out);
            if (out.has_radiance) {
                // per-sample accumulator: plain read-modify-write, the slot is owned by this ray
                float* acc = a.accum + ((int64_t)ray_id - a.id_base) * 3;
                acc[0] += out.radiance.r * a.inv_spi;
                acc[1] += out.radiance.g * a.inv_spi;
                acc[2] += out.radiance.b * a.inv_spi;
            }
        }

        // ---- append survivors / shadow rays: one atomic per wave and queue (replaces K9)
        {
            const unsigned long long m = __ballot(out.bounce);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                uint32_t pos0    = 0;
                if (lane == leader)
                    pos0 = atomicAdd(a.out_count, (uint32_t)__popcll(m));
                pos0 = __shfl(pos0, leader);
                if (out.bounce) {
                    const uint32_t o = pos0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    a.out.id[o] = ray_id;
                    a.out.ox[o] = out.b_org.x, a.out.oy[o] = out.b_org.y, a.out.oz[o] = out.b_org.z;
                    a.out.dx[o] = out.b_dir.x, a.out.dy[o] = out.b_dir.y, a.out.dz[o] = out.b_dir.z;
                    a.out.tmin[o]  = kRayOffset;
                    a.out.tmax[o]  = kFltMax;
                    a.out.flags[o] = IG_RAY_FLAG_BOUNCE;
                    a.out.rnd[o]   = out.b_rnd;
                    a.out.payload[0][o] = out.b_inv_pdf;
                    a.out.payload[1][o] = out.b_contrib.r;
                    a.out.payload[2][o] = out.b_contrib.g;
                    a.out.payload[3][o] = out.b_contrib.b;
                    a.out.payload[4][o] = (float)out.b_depth;
                    a.out.payload[5][o] = out.b_eta;
                }
            }
        }
        {
            const unsigned long long m = __ballot(out.shadow);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                uint32_t pos0    = 0;
                if (lane == leader)
                    pos0 = atomicAdd(a.sec_count, (uint32_t)__popcll(m));
                pos0 = __shfl(pos0, leader);
                if (out.shadow) {
                    const uint32_t o = pos0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    a.sec.id[o] = ray_id;
                    a.sec.ox[o] = out.s_org.x, a.sec.oy[o] = out.s_org.y, a.sec.oz[o] = out.s_org.z;
                    a.sec.dx[o] = out.s_dir.x, a.sec.dy[o] = out.s_dir.y, a.sec.dz[o] = out.s_dir.z;
                    a.sec.tmin[o] = kRayOffset;
                    a.sec.tmax[o] = out.s_tmax;
                    a.sec.cr[o] = out.s_col.r, a.sec.cg[o] = out.s_col.g, a.sec.cb[o] = out.s_col.b;
                }
            }
        }
    }
}

// Bookkeeping between bounce rounds, one thread: statistics (Statistics.h:57-64; BounceRayCount
// mapping_gpu.art:864, ShadowRayCount :857 -- real shadow rays here) and counter resets.
__global__ void k_round_end(QueueState* qs, int in_slot)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int out_slot = in_slot ^ 1;
        qs->bounce_rays += qs->primary_count[out_slot];
        qs->shadow_rays += qs->secondary_count;
        qs->primary_count[in_slot] = 0; // becomes the append target of the next round
        qs->work_counter[0]        = 0;
        qs->work_counter[1]        = 0;
        qs->work_counter[2]        = 0;
    }
}

// Runs after the shadow traversal of a round.
__global__ void k_secondary_end(QueueState* qs)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        qs->secondary_count = 0;
        qs->work_counter[2] = 0;
    }
}

// ---------------------------------------------------------------- k_resolve

// fb[pixel] += sum over samples of the per-sample accumulator, in sample order. The reference adds
// `color / spi` per event (driver/accumulator.art:4-30); the accumulators already hold those products.
__global__ void __launch_bounds__(256) k_resolve(const ResolveArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.pixels)
        return;
    const int64_t lp = a.first_local_pixel + i;
    const int x      = (int)(lp % a.width);
    const int y      = a.row_offset + (int)(lp / a.width) * a.row_stride;
    const float* src = a.accum + (size_t)i * a.spi * 3;
    float r = 0, g = 0, b = 0;
    for (int s = 0; s < a.spi; ++s) {
        r += src[s * 3 + 0];
        g += src[s * 3 + 1];
        b += src[s * 3 + 2];
    }
    float* dst = a.fb + ((size_t)y * a.width + x) * 3;
    dst[0] += r;
    dst[1] += g;
    dst[2] += b;
}

void launch_generate(const GenerateArgs& args, hipStream_t stream)
{
    const unsigned blocks = (args.n + 255u) / 256u;
    hipLaunchKernelGGL(k_generate, dim3(blocks ? blocks : 1u), dim3(256), 0, stream, args);
}

void launch_shade(const ShadeArgs& args, int grid_blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(k_shade, dim3((unsigned)grid_blocks), dim3(kShadeThreads), 0, stream, args);
}

void launch_round_end(QueueState* qs, int in_slot, hipStream_t stream) { hipLaunchKernelGGL(k_round_end, dim3(1), dim3(64), 0, stream, qs, in_slot); }
void launch_secondary_end(QueueState* qs, hipStream_t stream) { hipLaunchKernelGGL(k_secondary_end, dim3(1), dim3(64), 0, stream, qs); }

void launch_resolve(const ResolveArgs& args, hipStream_t stream)
{
    const unsigned blocks = (args.pixels + 255u) / 256u;
    hipLaunchKernelGGL(k_resolve, dim3(blocks ? blocks : 1u), dim3(256), 0, stream, args);
}

} // namespace igdev
