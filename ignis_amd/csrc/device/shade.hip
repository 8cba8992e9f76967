// shade.hip — ray generation, hit/miss shading + path-tracer callbacks, framebuffer resolve.
//
// Replaces the reference stages K1 (gpu_generate_rays, src/artic/driver/mapping_gpu.art:616-669),
// K3a-e (gpu_sort_primary, :409-502), K4 (gpu_hit_shade, :123-214), K5 (gpu_miss_shade, :237-274)
// and K9 (gpu_compact_primary, :686-711) with:
//   k_generate : one thread per camera sample, writes the SoA primary stream;
//   k_bin_*    : (scenes with the full BSDF library) the round's hits sorted by material, class-major, as one counting-sort pass
//                without global atomics; each material class of the shading kernels is then one run of the sorted index list;
//   k_shade    : surface reconstruction, emission/MIS, NEE shadow-ray emission, BSDF sampling + Russian roulette; survivors are
//                appended to the OTHER primary stream with one atomic per 256-ray window for both queues (M per-material
//                launches, host scan and compaction of the reference collapse into this kernel). The lean variant takes the hits
//                in stream order, the by-class variants their sorted runs, the one-for-all variant (IGD_SHADE_CLASSES=0, light
//                tracer, debug views, expressions) sorts 256 consecutive hits by material inside the workgroup;
//   k_resolve  : fixed-order sum of the per-sample accumulators into the framebuffer, so a given
//                seed reproduces the image bit for bit (the reference GPU path uses float atomics).
// Shading arithmetic follows src/artic/{core,bsdf,light,technique,camera} expression by expression
// (citations inline); transcendental functions come from include/ig_detmath.h.
#include <algorithm>

#include "shade_kernel.h"

namespace igdev {

// ---------------------------------------------------------------- k_generate

// permute_element (core/common.art:302-335)
IG_DEV uint32_t permute_element(uint32_t i, uint32_t l, uint32_t seed)
{
    uint32_t w = l - 1;
    if (w == 0)
        return 0;
    w |= w >> 1, w |= w >> 2, w |= w >> 4, w |= w >> 8, w |= w >> 16;
    do {
        i ^= seed;
        i *= 0xe170893du;
        i ^= seed >> 16;
        i ^= (i & w) >> 4;
        i ^= seed >> 8;
        i *= 0x0929eb3fu;
        i ^= seed >> 23;
        i ^= (i & w) >> 1;
        i *= 1 | seed >> 27;
        i *= 0x6935fa69u;
        i ^= (i & w) >> 11;
        i *= 0x74dcb303u;
        i ^= (i & w) >> 2;
        i *= 0x9e501cc3u;
        i ^= (i & w) >> 2;
        i *= 0xc860a3dfu;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    return (i + seed) % l;
}

// radical_inverse (sampler/pixel_sampler.art:37-54)
IG_DEV float radical_inverse(uint32_t index, uint32_t base)
{
    const uint32_t limit = 0xFFFFFFFFu / base - base;
    const float inv_base = 1.0f / (float)base;
    float inv_base_n     = 1;
    uint32_t reversed    = 0;
    while (index != 0 && reversed < limit) {
        const uint32_t next = index / base;
        reversed            = reversed * base + (index - next * base);
        inv_base_n *= inv_base;
        index = next;
    }
    return igm_min((float)reversed * inv_base_n, 1 - kFltEps);
}

// inverse_radical_inverse (:56-64)
IG_DEV uint32_t inverse_radical_inverse(uint32_t inv, uint32_t base, uint32_t digits)
{
    uint32_t index = 0;
    for (uint32_t i = 0; i < digits; ++i) {
        index = index * base + inv % base;
        inv /= base;
    }
    return index;
}

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
    // (ids stay below 2^31, igd_render checks: 32-bit arithmetic with the host's multipliers instead of 64-bit divisions)
    const int64_t lid     = a.first_local_id + i;
    const uint32_t uid    = (uint32_t)lid;
    const uint32_t it_u   = a.by_rays_per_iteration.div(uid);
    const int it_local    = (int)it_u; // multi-iteration call: which of its iterations
    const uint32_t within = uid - it_u * (uint32_t)a.rays_per_iteration;
    const uint32_t lpixel = a.by_spi.div(within);
    const int sample      = (int)(within - lpixel * (uint32_t)a.spi);
    const uint32_t lrow   = a.by_width.div(lpixel);
    const int x           = (int)(lpixel - lrow * (uint32_t)a.width);
    const int y           = a.row_offset + (int)lrow * a.row_stride;

    Tea rnd{ make_seed(sample, a.iteration + it_local, a.frame, x, y, a.seed), 1 };
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
        // make_camera_emitter (driver/emitter.art:6-16), the film's pixel sampler (sampler/pixel_sampler.art),
        // make_pixelcoord_from_xy (driver/camera.art:21-29), then Camera::generate_ray of the scene's camera
        float rx, ry;
        const uint32_t index = (uint32_t)((a.iteration + it_local) * a.spi + sample); // emitter.art:9
        if (a.cam.pixel_sampler == IG_PIXEL_SAMPLER_MJITT) {
            // make_mjitt_pixel_sampler(4, 4) (:13-34)
            const uint32_t seed = fnv_step(fnv_step(0x811C9DC5u, (uint32_t)x), (uint32_t)y);
            const float sx      = (float)permute_element(index % 4u, 4u, seed * 0xa511e9b3u);
            const float sy      = (float)permute_element(index / 4u, 4u, seed * 0x63d83595u);
            const float jx      = rnd.f32();
            const float jy      = rnd.f32();
            rx                  = (sx + (sy + jx) / 4.0f) / 4.0f;
            ry                  = (sy + (sx + jy) / 4.0f) / 4.0f;
        } else if (a.cam.pixel_sampler == IG_PIXEL_SAMPLER_HALTON) {
            // make_halton_pixel_sampler (:152-167) over the offset of setup_halton_pixel_sampler (:127-140); i32 arithmetic
            // wraps and the remainder is the signed one, as written there. The generator is not advanced.
            const uint32_t stride = a.halton_scale_x * a.halton_scale_y;
            int32_t offset        = 0;
            if (stride > 1) {
                const uint32_t dx = inverse_radical_inverse((uint32_t)x, 2, a.halton_exp_x);
                const uint32_t dy = inverse_radical_inverse((uint32_t)y, 3, a.halton_exp_y);
                const uint32_t s0 = (dx * (stride / a.halton_scale_x)) * (uint32_t)a.halton_inv_x;
                const uint32_t s1 = (dy * (stride / a.halton_scale_y)) * (uint32_t)a.halton_inv_y;
                offset            = (int32_t)(s0 + s1) % (int32_t)stride;
            }
            const uint32_t hindex = (uint32_t)offset + index * stride;
            rx                    = radical_inverse(hindex >> a.halton_exp_x, 2);
            ry                    = radical_inverse(hindex / a.halton_scale_y, 3);
        } else {
            // make_uniform_pixel_sampler (:4-10)
            rx = rnd.f32();
            ry = rnd.f32();
        }
        const float nx = 2 * ((float)x + rx) / ((float)a.width) - 1;
        const float ny = 1 - 2 * ((float)y + ry) / ((float)a.height);
        const f3 cdir  = f3{ a.cam.dir[0], a.cam.dir[1], a.cam.dir[2] };
        const f3 cup   = f3{ a.cam.up[0], a.cam.up[1], a.cam.up[2] };
        m33 view;
        view.c0 = normalize3(cross3(cdir, cup));
        view.c1 = cup;
        view.c2 = cdir;
        org     = f3{ a.cam.eye[0], a.cam.eye[1], a.cam.eye[2] };
        tmin    = a.cam.near_clip;
        tmax    = a.cam.far_clip;
        flags   = IG_RAY_FLAG_CAMERA;
        if (a.cam.type == IG_CAMERA_ORTHOGONAL) {
            // make_orthogonal_camera (camera/orthogonal.art:19-22)
            org = mul33(view, f3{ a.sx * nx, a.sy * ny, 0 }) + org;
            dir = cdir;
        } else if (a.cam.type == IG_CAMERA_FISHLENS) {
            // compute_d of make_fishlens_camera (camera/fishlens.art:39-53), fov = pi; (sx, sy) = (xasp, yasp)
            const float fx    = nx * a.sx;
            const float fy    = ny * a.sy;
            const float r     = igm_sqrt(fx * fx + fy * fy);
            const float theta = r * kPi / 2;
            const float sT    = igm_sin(theta);
            const float cT    = igm_cos(theta);
            const float sP    = r < kFltEps ? 0 : fy / r;
            const float cP    = r < kFltEps ? 0 : fx / r;
            dir               = mul33(view, f3{ sT * cP, sT * sP, cT });
            if (a.cam.fisheye_mask && r > 1) {
                // no ray for this sample: the zero ray of gpu_generate_rays (mapping_gpu.art:655-658);
                // it cannot hit anything and shade_vertex drops it
                org  = f3{ 0, 0, 0 };
                dir  = f3{ 0, 0, 0 };
                tmin = 0;
                tmax = 0;
                flags = 0;
            }
        } else {
            // make_perspective_camera (camera/perspective.art:29-42)
            dir = normalize3(mul33(view, f3{ a.sx * nx, a.sy * ny, 1 }));
            if (a.cam.aperture_radius > kFltEps) {
                // gen_ray of make_perspective_dof_camera (camera/perspective.art:73-84)
                const f3 focus = dir * a.cam.focal_length;
                const float u0 = rnd.f32();
                const float u1 = rnd.f32();
                // square_to_concentric_disk (core/warp.art:2-22)
                const float ca = 2 * u0 - 1;
                const float cb = 2 * u1 - 1;
                float ax = 0, ay = 0;
                if (ca == 0 && cb == 0) {
                } else if (ca * ca > cb * cb) {
                    const float phi = (kPi / 4) * safe_div(cb, ca);
                    ax              = igm_cos(phi) * ca;
                    ay              = igm_sin(phi) * ca;
                } else {
                    const float phi = (kPi / 2) - (kPi / 4) * safe_div(ca, cb);
                    ax              = igm_cos(phi) * cb;
                    ay              = igm_sin(phi) * cb;
                }
                const f3 ap = mul33(view, f3{ ax * a.cam.aperture_radius, ay * a.cam.aperture_radius, 0 });
                org         = org + ap;
                dir         = normalize3(focus - ap);
            }
        }
    }

    a.out.rayB[i] = make_float4(dir.x, dir.y, dir.z, tmax);
    if (a.accum_clear)
        a.accum_clear[i] = make_float4(0, 0, 0, 0);
    if (a.compact)
        return; // (CameraStream::compact: origin, near clip, flags, depth and the generator's counter are the same for every ray)
    a.out.rayA[i] = make_float4(org.x, org.y, org.z, tmin);
    // init_pt_raypayload (technique/pathtracer.art:33-38): inv_pdf 0, contrib white, depth 1, eta 1
    // (kStreamCamera: the payload of init_pt_raypayload — inv_pdf 0, contrib white, depth 1, eta 1, technique/pathtracer.art:33-38 — is
    // the same for every camera ray: the readers supply it, the pay / eta columns are not written)
    a.out.meta[i] = make_int4((int32_t)lid, (int32_t)flags, (int32_t)rnd.counter, 1);
}

// make_lt_emitter (technique/lighttracer.art:35-62): the light selector at position 0, Light::sample_emission, payload
// (contrib = intensity * |cos| / light_pdf, depth 1, eta 1). A sample without a ray gets the zero ray, which can only miss.
__global__ void __launch_bounds__(256) k_generate_light(const GenerateLightArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        *a.out_count = a.n;
        atomicAdd(&a.qs->camera_rays, (unsigned long long)a.n);
    }
    if (i >= a.n)
        return;
    const int64_t lid    = a.first_local_id + i;
    const int it_local   = (int)(lid / a.rays_per_iteration);
    const int64_t within = lid % a.rays_per_iteration;
    const int sample     = (int)(within % a.spi);
    const int64_t lpixel = within / a.spi;
    const int x          = (int)(lpixel % a.width);
    const int y          = (int)(lpixel / a.width);
    Tea rnd{ make_seed(sample, a.iteration + it_local, a.frame, x, y, a.seed), 1 };

    f3 org{ 0, 0, 0 }, dir{ 0, 0, 0 };
    float tmin = 0, tmax = 0;
    uint32_t flags = 0;
    Col contrib{ 0, 0, 0 };
    int li_used = 0;
    const DevScene& sc = a.scene;
    if (sc.light_count > 0) {
        float light_pdf;
        const int li = select_light<true>(sc, rnd, f3{ 0, 0, 0 }, light_pdf);
        li_used      = li;
        EmissionSample es;
        if (sample_emission(sc, sc.lights[li], rnd, es)) {
            org     = es.pos;
            dir     = es.dir;
            tmin    = (uint32_t)li < sc.infinite_light_count ? 0.0f : kRayOffset;
            tmax    = kFltMax;
            flags   = IG_RAY_FLAG_LIGHT;
            contrib = es.intensity * safe_div(igm_abs(es.cos), light_pdf * 1.0f);
        }
    }
    a.out.rayA[i] = make_float4(org.x, org.y, org.z, tmin);
    a.out.rayB[i] = make_float4(dir.x, dir.y, dir.z, tmax);
    a.out.meta[i] = make_int4((int32_t)lid, (int32_t)flags, (int32_t)rnd.counter, 1);
    a.out.pay[i]  = make_float4(a.ppm ? (float)li_used : 0.0f, contrib.r, contrib.g, contrib.b); // (kStreamLight: eta 1 is the readers')
}

// Bookkeeping between bounce rounds, one thread: statistics (Statistics.h:57-64; BounceRayCount
// mapping_gpu.art:864, ShadowRayCount :857 -- real shadow rays here) and counter resets.
__global__ void k_round_end(QueueState* qs, int in_slot)
{
    if (threadIdx.x < kWorkShards && blockIdx.x == 0)
        for (int k = 0; k < 6; ++k)
            qs->work[k].w[threadIdx.x][0] = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int out_slot = in_slot ^ 1;
        qs->bounce_rays += qs->q[out_slot].primary;
        qs->shadow_rays += qs->q[out_slot].secondary;
        qs->q[in_slot].primary = 0; // becomes the append target of the next round (its .secondary is already 0)
        qs->deep_total += qs->deep_count;
        qs->deep_count             = 0;
    }
}

// Runs after the shadow traversal of a round.
// `mirror` (optional) is host memory mapped into the device: the queue state of the finished round is written
// there by this kernel, so the host never needs a copy-engine transfer between two rounds (a 64-byte
// hipMemcpyAsync D2H put ~155 us of engine hand-over between the shadow traversal and the next round).
__global__ void k_secondary_end(QueueState* qs, int slot, QueueState* mirror)
{
    if (threadIdx.x < kWorkShards && blockIdx.x == 0)
        qs->work[2].w[threadIdx.x][0] = 0, qs->work[3].w[threadIdx.x][0] = 0, qs->work[5].w[threadIdx.x][0] = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        qs->q[slot].secondary = 0;
        qs->deep_total += qs->deep_count;
        qs->deep_count      = 0;
    }
    if (mirror && blockIdx.x == 0) {
        // (everything but the work counters, a word per thread and trip)
        __syncthreads();
        const uint32_t* src = reinterpret_cast<const uint32_t*>(qs);
        uint32_t* dst       = reinterpret_cast<uint32_t*>(mirror);
        for (uint32_t i = threadIdx.x; i < kQueueStateHead / 4; i += blockDim.x)
            dst[i] = src[i];
        __threadfence_system();
    }
}

// ---------------------------------------------------------------- k_info

// wrap_infobuffer_renderer.on_hit (technique/internal/infobuffer.art:9-25): for the camera rays of iteration 0 the shading
// normal and the saturated BSDF albedo of the first hit, times 1 / spi like every splat (driver/accumulator.art:4-30)
__global__ void __launch_bounds__(256) k_info(const InfoArgs a)
{
    const uint32_t n = *a.count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int4 meta   = a.in.meta[i];
        const float4 hit  = a.in.hit[i];
        const int ent     = (int)igm_bits(hit.x);
        float4 nrm = make_float4(0, 0, 0, 0), alb = make_float4(0, 0, 0, 0);
        if (ent >= 0 && ((uint32_t)meta.y & IG_RAY_FLAG_CAMERA)) {
            const float4 ra = a.in.rayA[i], rb = a.in.rayB[i];
            const f3 org{ ra.x, ra.y, ra.z }, dir{ rb.x, rb.y, rb.z };
            const ig_material& mat = a.scene.materials[a.scene.entity_material[ent]];
            const Surf surf        = surface_element<true>(a.scene, ent, (int)igm_bits(hit.y), org, dir, hit.z, hit.w, a.in.hit_v[i]);
            const BsdfCtx<true, true, true> bsdf(a.scene, mat, surf, dir, std::true_type{});
            const Col al = bsdf.albedo(-dir);
            const f3 N   = surf.local.c2;
            nrm          = make_float4(N.x * a.inv_spi, N.y * a.inv_spi, N.z * a.inv_spi, 0);
            alb          = make_float4(igm_min(al.r, 1.0f) * a.inv_spi, igm_min(al.g, 1.0f) * a.inv_spi, igm_min(al.b, 1.0f) * a.inv_spi, 0);
        }
        const int64_t slot = (int64_t)meta.x - a.id_base;
        a.normals[slot]    = nrm;
        a.albedo[slot]     = alb;
    }
}

void launch_info(const InfoArgs& args, int grid_blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(k_info, dim3((unsigned)grid_blocks), dim3(256), 0, stream, args);
}

// ---------------------------------------------------------------- k_resolve

// fb[pixel] += sum over samples of the per-sample accumulator, in sample order. The reference adds
// `color / spi` per event (driver/accumulator.art:4-30); the accumulators already hold those products.
// One workgroup handles tiles of kResolveFloat4 / spi pixels: the tile's accumulators are one contiguous run, loaded
// coalesced into LDS (a thread reading its pixel's spi slots directly strides 16 * spi bytes between lanes and fetched
// every line three times from HBM), then one thread per pixel adds its samples in order, iteration after iteration.
constexpr int kResolveFloat4 = 2048; // 32 KiB of LDS

__global__ void __launch_bounds__(256) k_resolve(const ResolveArgs a)
{
    __shared__ float4 s_acc[kResolveFloat4];
    const int tid       = threadIdx.x;
    const uint32_t spi  = (uint32_t)a.spi;
    const uint32_t tile = kResolveFloat4 / spi < 256u ? (kResolveFloat4 / spi > 0u ? kResolveFloat4 / spi : 1u) : 256u; // pixels per pass
    // pixels this launch covers: a whole iteration's (multi-iteration chunks) or the chunk's virtual pixels
    const uint32_t n_pix  = a.iterations > 1 ? a.local_pixels : a.pixels;
    const uint32_t n_iter = a.iterations > 1 ? a.iterations : 1u;
    const uint32_t tiles  = (n_pix + tile - 1) / tile;
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint32_t p0 = t * tile;
        const uint32_t np = n_pix - p0 < tile ? n_pix - p0 : tile;
        // destination of this thread's pixel
        float* dst = nullptr;
        float fr = 0, fg = 0, fbb = 0;
        if ((uint32_t)tid < np) {
            const int64_t lp = a.iterations > 1 ? (int64_t)(p0 + tid) : (a.first_local_pixel + p0 + tid) % a.local_pixels;
            const int x      = (int)(lp % a.width);
            const int y      = a.row_offset + (int)(lp / a.width) * a.row_stride;
            dst              = a.fb + ((size_t)y * a.width + x) * 3;
            fr = dst[0], fg = dst[1], fbb = dst[2];
        }
        for (uint32_t it = 0; it < n_iter; ++it) {
            const float4* src = a.accum + ((size_t)it * a.local_pixels * (a.iterations > 1 ? 1u : 0u) + p0) * spi;
            if (spi <= (uint32_t)kResolveFloat4) {
                for (uint32_t e = (uint32_t)tid; e < np * spi; e += 256u)
                    s_acc[e] = src[e];
                __syncthreads();
                if ((uint32_t)tid < np) {
                    float r = 0, g = 0, b = 0;
                    for (uint32_t k = 0; k < spi; ++k) {
                        const float4 v = s_acc[(uint32_t)tid * spi + k];
                        r += v.x, g += v.y, b += v.z;
                    }
                    fr += r, fg += g, fbb += b;
                }
                __syncthreads();
            } else if ((uint32_t)tid < np) { // more samples per pixel than the staging buffer holds: read them directly
                float r = 0, g = 0, b = 0;
                for (uint32_t k = 0; k < spi; ++k) {
                    const float4 v = src[(size_t)tid * spi + k];
                    r += v.x, g += v.y, b += v.z;
                }
                fr += r, fg += g, fbb += b;
            }
        }
        if (dst)
            dst[0] = fr, dst[1] = fg, dst[2] = fbb;
    }
}

// extended_gcd (sampler/pixel_sampler.art:77-85)
static void extended_gcd(uint32_t a, uint32_t b, int32_t& x, int32_t& y)
{
    if (b == 0) {
        x = 1, y = 0;
        return;
    }
    int32_t xx, yy;
    extended_gcd(b, a % b, xx, yy);
    x = yy;
    y = (int32_t)((uint32_t)xx - (a / b) * (uint32_t)yy);
}

void launch_generate(const GenerateArgs& in, hipStream_t stream)
{
    GenerateArgs args = in;
    args.by_rays_per_iteration = FastDiv::make((uint32_t)args.rays_per_iteration);
    args.by_spi                = FastDiv::make((uint32_t)args.spi);
    args.by_width              = FastDiv::make((uint32_t)args.width);
    if (args.cam.pixel_sampler == IG_PIXEL_SAMPLER_HALTON) {
        // compute_halton_base_info + multiplicative_inverse of setup_halton_pixel_sampler (:66-75,87-90,107-111)
        args.halton_scale_x = 1, args.halton_exp_x = 0;
        while (args.halton_scale_x < (uint32_t)args.width)
            args.halton_scale_x *= 2, ++args.halton_exp_x;
        args.halton_scale_y = 1, args.halton_exp_y = 0;
        while (args.halton_scale_y < (uint32_t)args.height)
            args.halton_scale_y *= 3, ++args.halton_exp_y;
        int32_t x, y;
        extended_gcd(args.halton_scale_x, args.halton_scale_y, x, y);
        args.halton_inv_x = x % (int32_t)args.halton_scale_y;
        extended_gcd(args.halton_scale_y, args.halton_scale_x, x, y);
        args.halton_inv_y = x % (int32_t)args.halton_scale_x;
    }
    const unsigned blocks = (args.n + 255u) / 256u;
    hipLaunchKernelGGL(k_generate, dim3(blocks ? blocks : 1u), dim3(256), 0, stream, args);
}

template __global__ void k_shade<false>(const ShadeArgs);
template __global__ void k_shade<true>(const ShadeArgs);
template __global__ void k_shade<true, true, true>(const ShadeArgs);
template __global__ void k_shade<true, false, true>(const ShadeArgs);
template __global__ void k_shade<true, false, true, true>(const ShadeArgs);
template __global__ void k_shade<true, false, false, false, 0, kClassBasic>(const ShadeArgs);
template __global__ void k_shade<true, false, false, false, 0, kClassPrincipled>(const ShadeArgs);
template __global__ void k_shade<true, false, false, false, 0, kClassCoated>(const ShadeArgs);
template __global__ void k_shade<true, false, false, false, 0, kClassBlend>(const ShadeArgs);

// ---------------------------------------------------------------- k_bin_*: the round's hits sorted by material
//
// A counting sort in the shape of one radix-sort digit pass, without a single global atomic (round 5's first version reserved
// a window's share of a bin with one atomic per window and bin: 160 K same-address atomics per 66 M hits bounded it at 0.38 ms):
// workgroup c of G owns the contiguous chunk c of the hits. k_bin_count: the chunk's histogram -> wg_hist[bin][c] (and the key byte
// of every ray); k_bin_prefix: per bin, the exclusive prefix over the G chunks and the bin's total; k_bin_scan: the bins' first
// slots in class-major order and the classes' runs; k_bin_scatter: the chunk's rays into their slots from LDS cursors that start at
// first[bin] + prefix[bin][c]. Inside a bin the rays of a chunk stay together and chunks follow each other in stream order: the
// shading kernel's gathers walk the stream forwards.
IG_DEV uint32_t bin_chunk_len(uint32_t n, uint32_t grid) { return (((n + grid - 1u) / grid) + 255u) & ~255u; }

__global__ void __launch_bounds__(256) k_bin_count(const BinSortArgs a)
{
    __shared__ uint32_t s_hist[kSortBins];
    const uint32_t tid = threadIdx.x;
    s_hist[tid]        = 0;
    __syncthreads();
    const uint32_t n   = *a.count;
    const uint32_t M   = a.material_count;
    const uint32_t len = bin_chunk_len(n, gridDim.x);
    const uint32_t lo = blockIdx.x * len, hi = lo + len < n ? lo + len : n;
    for (uint32_t i = lo + tid; i < hi; i += 256u) {
        int ent = (int)igm_bits(a.hit[i].x), prim_;
        if (a.hit_pack)
            unpack_hit_ids(a.hit_pack, igm_bits(a.hit[i].x), ent, prim_);
        const uint32_t key = ent < 0 ? M : (uint32_t)a.entity_material[ent];
        a.keys[i]          = (uint8_t)key;
        // one LDS atomic for the lanes that share the first lane's key (neighbouring rays hit the same material more often than not;
        // 64 same-address LDS atomics serialise), one each for the others
        const unsigned long long lm   = __ballot(true);
        const uint32_t lead_key       = (uint32_t)__builtin_amdgcn_readlane((int)key, __builtin_ctzll(lm));
        const unsigned long long same = __ballot(key == lead_key);
        if (key != lead_key)
            atomicAdd(&s_hist[key], 1u);
        else if ((tid & 63u) == (uint32_t)__builtin_ctzll(same))
            atomicAdd(&s_hist[lead_key], (uint32_t)__popcll(same));
    }
    __syncthreads();
    if (tid <= M)
        a.wg_hist[(size_t)tid * gridDim.x + blockIdx.x] = s_hist[tid];
}

// one workgroup per bin: wg_hist[bin][0 .. G) -> its exclusive prefix, the total -> state[bin]
__global__ void __launch_bounds__(256) k_bin_prefix(const BinSortArgs a, uint32_t grid)
{
    __shared__ uint32_t s_part[256];
    const uint32_t tid = threadIdx.x, bin = blockIdx.x;
    uint32_t* row      = a.wg_hist + (size_t)bin * grid;
    const uint32_t per = (grid + 255u) / 256u; // entries per thread, consecutive
    uint32_t sum       = 0;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t c = tid * per + k;
        sum += c < grid ? row[c] : 0u;
    }
    s_part[tid] = sum;
    __syncthreads();
    // Hillis-Steele over the 256 partial sums
    for (uint32_t off = 1; off < 256u; off <<= 1) {
        const uint32_t v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum; // exclusive
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t c = tid * per + k;
        if (c < grid) {
            const uint32_t v = row[c];
            row[c]           = run;
            run += v;
        }
    }
    if (tid == 255u)
        a.state[bin] = s_part[255];
}

// one workgroup: the bins' first slots in class-major order (bin_order) and each class's {first, count}
__global__ void __launch_bounds__(256) k_bin_scan(const BinSortArgs a)
{
    __shared__ uint32_t s_cnt[kSortBins], s_first[kSortBins];
    const uint32_t tid  = threadIdx.x;
    const uint32_t bins = a.material_count + 1;
    const uint32_t bin  = tid < bins ? a.bin_order[tid] : 0u;
    s_cnt[tid]          = tid < bins ? a.state[bin] : 0u;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0, cls_first[kSortClasses] = {}, cls_count[kSortClasses] = {};
        int last = -1;
        for (uint32_t k = 0; k < bins; ++k) {
            const int c = a.bin_class[a.bin_order[k]];
            s_first[k]  = run;
            if (c >= kSortClasses)
                continue; // a bin no kernel shades (kSortDeadBin: the misses of a scene without environment lights): no slots
            if (c != last)
                cls_first[c] = run, last = c;
            cls_count[c] += s_cnt[k];
            run += s_cnt[k];
        }
        for (int c = 0; c < kSortClasses; ++c) {
            a.state[3 * kSortBins + 2 * c]     = cls_first[c];
            a.state[3 * kSortBins + 2 * c + 1] = cls_count[c];
        }
    }
    __syncthreads();
    if (tid < bins)
        a.state[kSortBins + bin] = s_first[tid];
}

__global__ void __launch_bounds__(256) k_bin_scatter(const BinSortArgs a)
{
    __shared__ uint32_t s_cursor[kSortBins];
    const uint32_t tid = threadIdx.x;
    const uint32_t n   = *a.count;
    const uint32_t M   = a.material_count;
    s_cursor[tid]      = tid <= M ? a.state[kSortBins + tid] + a.wg_hist[(size_t)tid * gridDim.x + blockIdx.x] : 0u;
    const bool dead    = tid <= M && a.bin_class[tid] >= kSortClasses;
    __shared__ uint8_t s_dead[kSortBins];
    s_dead[tid] = dead ? 1 : 0;
    __syncthreads();
    const uint32_t len = bin_chunk_len(n, gridDim.x);
    const uint32_t lo = blockIdx.x * len, hi = lo + len < n ? lo + len : n;
    const uint32_t lane_id = tid & 63u;
    for (uint32_t i = lo + tid; i < hi; i += 256u) {
        const uint32_t key = a.keys[i];
        const bool live    = !s_dead[key]; // (a dead bin's rays are nobody's: no slot)
        const unsigned long long lm = __ballot(live);
        if (lm) {
            // the lanes that share the first live lane's key take consecutive slots from ONE LDS atomic, the others one each
            const uint32_t lead_key       = (uint32_t)__builtin_amdgcn_readlane((int)key, __builtin_ctzll(lm));
            const unsigned long long same = __ballot(live && key == lead_key);
            uint32_t base                 = 0;
            if (lane_id == (uint32_t)__builtin_ctzll(same))
                base = atomicAdd(&s_cursor[lead_key], (uint32_t)__popcll(same));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(same));
            if (live && key == lead_key)
                a.sort_idx[base + (uint32_t)__popcll(same & ((1ull << lane_id) - 1ull))] = i;
            else if (live)
                a.sort_idx[atomicAdd(&s_cursor[key], 1u)] = i;
        }
    }
}

void launch_bin_sort(const BinSortArgs& args, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(k_bin_count, dim3((unsigned)grid), dim3(256), 0, stream, args);
    hipLaunchKernelGGL(k_bin_prefix, dim3(args.material_count + 1u), dim3(256), 0, stream, args, (uint32_t)grid);
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(256), 0, stream, args);
    hipLaunchKernelGGL(k_bin_scatter, dim3((unsigned)grid), dim3(256), 0, stream, args);
}

void launch_generate_light(const GenerateLightArgs& args, hipStream_t stream)
{
    const unsigned blocks = (args.n + 255u) / 256u;
    hipLaunchKernelGGL(k_generate_light, dim3(blocks ? blocks : 1u), dim3(256), 0, stream, args);
}

// classes: which material classes the scene has (bit 0 basic — always launched: the misses are its —, 1 principled, 2 coated,
// 3 blend); 0: the one instantiation with every model
void launch_shade(const ShadeArgs& args, int grid_blocks, bool full_bsdfs, hipStream_t stream, uint32_t classes)
{
    if (args.scene.tech.type == IG_TECHNIQUE_LIGHTTRACER)
        hipLaunchKernelGGL((k_shade<true, false, true, true>), dim3((unsigned)grid_blocks), dim3(kShadeThreads), 0, stream, args);
    else if (args.scene.tech.type == IG_TECHNIQUE_DEBUG || args.scene.tech.type == IG_TECHNIQUE_WIREFRAME) // (the tail kernels never see these techniques)
        hipLaunchKernelGGL((k_shade<true, true, true>), dim3((unsigned)grid_blocks), dim3(kShadeThreads), 0, stream, args);
    else if (args.scene.expr_code) // materials with shading expressions: the instantiation with the interpreter (no tail kernels either)
        hipLaunchKernelGGL((k_shade<true, false, true>), dim3((unsigned)grid_blocks), dim3(kShadeThreads), 0, stream, args);
    else if (full_bsdfs && classes != 0) {
        // (the caller has run launch_bin_sort on this round's hits: args.sort_idx, args.cls_range = the sort's state words)
        ShadeArgs ca       = args;
        const uint32_t* cr = args.cls_range + 3 * kSortBins;
        ca.cls_range       = cr;
        hipLaunchKernelGGL((k_shade<true, false, false, false, 0, kClassBasic>), dim3((unsigned)grid_blocks), dim3(kShadeThreads), 0, stream, ca);
        ca.cls_range = cr + 2;
        if (classes & 2u)
            hipLaunchKernelGGL((k_shade<true, false, false, false, 0, kClassPrincipled>), dim3((unsigned)grid_blocks), dim3(kShadeThreads), 0, stream, ca);
        ca.cls_range = cr + 4;
        if (classes & 4u)
            hipLaunchKernelGGL((k_shade<true, false, false, false, 0, kClassCoated>), dim3((unsigned)grid_blocks), dim3(kShadeThreads), 0, stream, ca);
        ca.cls_range = cr + 6;
        if (classes & 8u)
            hipLaunchKernelGGL((k_shade<true, false, false, false, 0, kClassBlend>), dim3((unsigned)grid_blocks), dim3(kShadeThreads), 0, stream, ca);
    } else if (full_bsdfs)
        hipLaunchKernelGGL((k_shade<true>), dim3((unsigned)grid_blocks), dim3(kShadeThreads), 0, stream, args);
    else
        hipLaunchKernelGGL((k_shade<false>), dim3((unsigned)grid_blocks), dim3(kShadeThreads), 0, stream, args);
}

void launch_round_end(QueueState* qs, int in_slot, hipStream_t stream) { hipLaunchKernelGGL(k_round_end, dim3(1), dim3(64), 0, stream, qs, in_slot); }
void launch_secondary_end(QueueState* qs, int slot, QueueState* mirror, hipStream_t stream)
{
    hipLaunchKernelGGL(k_secondary_end, dim3(1), dim3(64), 0, stream, qs, slot, mirror);
}

// Moves the surviving paths (the columns the tail kernel reads) out of a primary stream. A compact camera stream (kernels.h CameraStream)
// is written out in full: the tail kernel reads rayA and meta like any other stream's.
__global__ void __launch_bounds__(256) k_copy_paths(PrimaryCols src, PrimaryCols dst, const uint32_t* __restrict__ count, CameraStream cam)
{
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        dst.rayA[i] = cam.compact ? cam.rayA : src.rayA[i];
        dst.rayB[i] = src.rayB[i];
        dst.meta[i] = cam.compact ? make_int4((int32_t)(cam.first_id + i), (int32_t)IG_RAY_FLAG_CAMERA, (int32_t)cam.rnd_counter, 1) : src.meta[i];
        dst.pay[i]  = src.pay[i]; // (eta travels in meta.y or is the constant 1: kernels.h kStream*)
    }
}

void launch_copy_paths(const PrimaryCols& src, const PrimaryCols& dst, const uint32_t* count, uint32_t max_count, const CameraStream& cam, hipStream_t stream)
{
    const unsigned blocks = std::min(4096u, (max_count + 255u) / 256u);
    hipLaunchKernelGGL(k_copy_paths, dim3(blocks ? blocks : 1u), dim3(256), 0, stream, src, dst, count, cam);
}

void launch_resolve(const ResolveArgs& args, hipStream_t stream)
{
    const unsigned n_pix = args.iterations > 1 ? args.local_pixels : args.pixels;
    const unsigned spi   = (unsigned)std::max(1, args.spi);
    const unsigned tile  = std::min(256u, std::max(1u, (unsigned)kResolveFloat4 / spi));
    const unsigned tiles = (n_pix + tile - 1) / tile;
    hipLaunchKernelGGL(k_resolve, dim3(std::min(tiles ? tiles : 1u, 16384u)), dim3(256), 0, stream, args);
}

} // namespace igdev
