// raysort.hip — rays of a traversal launch ordered in space (round 6).
//
// Not a stage of the reference: its wavefront sorts hits by material for the shading (gpu_sort_primary, mapping_gpu.art:409-502) and
// traverses rays in arrival order. On a BVH that outgrows the L2s every node visit of a bounce ray in arrival order is a cache miss of
// its own (profiles/r05_traffic_standin_divergent.json: 472 B of L2-miss traffic per ray, lane utilisation 0.49). Here the rays k_shade
// appended get a key — the octant of the direction and the Morton code of the origin inside the scene's box — and the traversal
// launch takes them in key order (TraverseArgs::sort_idx): a wave's 64 rays start in one neighbourhood and walk the children in the
// same order. Per-ray results and the work counters are sums over rays, so hits, radiance and counters do not depend on the order.
//
// The sort is k_bin_*'s counting sort (shade.hip) as the digit pass of an LSD radix sort, 8 bits at a time, without a global atomic:
// workgroup c of G owns the contiguous chunk c of the stream in its current order; k_rs_count: the chunk's histogram of the pass's
// digit -> wg_hist[digit][c] (the first pass makes the keys on the way); k_rs_prefix: per digit the exclusive prefix over the chunks;
// k_rs_scatter: (key, ray index) pairs into their slots from LDS cursors. A pass keeps the order of the one before between chunks and
// between the windows of a chunk, not inside a window of 256 (slots come from LDS atomics): the result is sorted by the high digits
// and nearly sorted by the low ones, which is all a scheduling order needs.
#include "kernels.h"
#include "dev_math.h"

namespace igdev {

IG_DEV uint32_t rs_chunk_len(uint32_t n, uint32_t grid) { return (((n + grid - 1u) / grid) + 255u) & ~255u; }

// bits 0, 3, 6, ... of the result = the low ten bits of v
IG_DEV uint32_t rs_spread3(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// bits 0, 2, 4, ... of the result = the low sixteen bits of v
IG_DEV uint32_t rs_spread2(uint32_t v)
{
    v &= 0xFFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

IG_DEV uint32_t rs_key(const RaySortArgs& a, const float4 ra, const float4 rb)
{
    const float top = (float)((1u << a.cell_bits) - 1u);
    // (a NaN origin compares false twice and lands in cell 0: any key is a valid key)
    const float fx = (ra.x - a.box_min[0]) * a.box_scale[0], fy = (ra.y - a.box_min[1]) * a.box_scale[1], fz = (ra.z - a.box_min[2]) * a.box_scale[2];
    const uint32_t qx = (uint32_t)(fx > 0 ? (fx < top ? fx : top) : 0.0f), qy = (uint32_t)(fy > 0 ? (fy < top ? fy : top) : 0.0f), qz = (uint32_t)(fz > 0 ? (fz < top ? fz : top) : 0.0f);
    const uint32_t morton = rs_spread3(qx) | (rs_spread3(qy) << 1) | (rs_spread3(qz) << 2);
    if (a.dir_bits) {
        // direction-major: the direction's cell on the octahedral map of the sphere (2 x dir_bits bits, Morton order), then the origin's cell:
        // a wave's rays are neighbours with nearly one direction, the bundle an orthographic camera would shoot
        const float l1 = igm_abs(rb.x) + igm_abs(rb.y) + igm_abs(rb.z);
        const float s  = l1 > 0 ? 1.0f / l1 : 0.0f;
        float px = rb.x * s, py = rb.y * s;
        if (rb.z < 0) {
            const float ox = (1.0f - igm_abs(py)) * (px < 0 ? -1.0f : 1.0f), oy = (1.0f - igm_abs(px)) * (py < 0 ? -1.0f : 1.0f);
            px = ox, py = oy;
        }
        const float dtop = (float)((1u << a.dir_bits) - 1u), dn = (float)(1u << a.dir_bits) * 0.5f;
        const float du = (px + 1.0f) * dn, dv = (py + 1.0f) * dn;
        const uint32_t qu = (uint32_t)(du > 0 ? (du < dtop ? du : dtop) : 0.0f), qv = (uint32_t)(dv > 0 ? (dv < dtop ? dv : dtop) : 0.0f);
        const uint32_t dcell = rs_spread2(qu) | (rs_spread2(qv) << 1);
        return a.octant_low ? (morton << (2u * a.dir_bits)) | dcell : (dcell << (3u * a.cell_bits)) | morton;
    }
    const uint32_t octant = (rb.x < 0 ? 1u : 0u) | (rb.y < 0 ? 2u : 0u) | (rb.z < 0 ? 4u : 0u);
    return a.octant_low ? (morton << 3) | octant : (octant << (3u * a.cell_bits)) | morton;
}

template <bool FIRST>
__global__ void __launch_bounds__(256) k_rs_count(const RaySortArgs a, const uint32_t* __restrict__ keys_in, uint32_t shift)
{
    __shared__ uint32_t s_hist[256];
    const uint32_t tid = threadIdx.x;
    s_hist[tid]        = 0;
    __syncthreads();
    const uint32_t n   = *a.count;
    const uint32_t len = rs_chunk_len(n, gridDim.x);
    const uint32_t lo = blockIdx.x * len, hi = lo + len < n ? lo + len : n;
    for (uint32_t i = lo + tid; i < hi; i += 256u) {
        uint32_t key;
        if (FIRST) {
            key       = rs_key(a, a.rayA[i], a.rayB[i]);
            a.keys[0][i] = key;
        } else {
            key = keys_in[i];
        }
        atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    a.wg_hist[(size_t)tid * gridDim.x + blockIdx.x] = s_hist[tid];
}

// one workgroup per digit value: wg_hist[digit][0 .. G) -> its exclusive prefix, the total -> state[digit]
__global__ void __launch_bounds__(256) k_rs_prefix(const RaySortArgs a, uint32_t grid)
{
    __shared__ uint32_t s_part[256];
    const uint32_t tid = threadIdx.x, bin = blockIdx.x;
    uint32_t* row      = a.wg_hist + (size_t)bin * grid;
    const uint32_t per = (grid + 255u) / 256u;
    uint32_t sum       = 0;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t c = tid * per + k;
        sum += c < grid ? row[c] : 0u;
    }
    s_part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 256u; off <<= 1) {
        const uint32_t v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;
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

// FIRST: the pairs are (keys[0][i], i); LAST: only the ray indices leave (nobody reads the keys again)
template <bool FIRST, bool LAST>
__global__ void __launch_bounds__(256) k_rs_scatter(const RaySortArgs a, const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in, uint32_t* __restrict__ keys_out,
                                                    uint32_t* __restrict__ idx_out, uint32_t shift)
{
    __shared__ uint32_t s_cursor[256];
    const uint32_t tid = threadIdx.x;
    // the digits' first slots: the exclusive scan of their totals (every workgroup for itself: 256 words)
    const uint32_t total = a.state[tid];
    s_cursor[tid]        = total;
    __syncthreads();
    for (uint32_t off = 1; off < 256u; off <<= 1) {
        const uint32_t v = tid >= off ? s_cursor[tid - off] : 0u;
        __syncthreads();
        s_cursor[tid] += v;
        __syncthreads();
    }
    const uint32_t first = s_cursor[tid] - total;
    __syncthreads();
    s_cursor[tid] = first + a.wg_hist[(size_t)tid * gridDim.x + blockIdx.x];
    __syncthreads();
    const uint32_t n   = *a.count;
    const uint32_t len = rs_chunk_len(n, gridDim.x);
    const uint32_t lo = blockIdx.x * len, hi = lo + len < n ? lo + len : n;
    for (uint32_t i = lo + tid; i < hi; i += 256u) {
        const uint32_t key  = keys_in[i];
        const uint32_t src  = FIRST ? i : idx_in[i];
        const uint32_t slot = atomicAdd(&s_cursor[(key >> shift) & 255u], 1u);
        if (!LAST)
            keys_out[slot] = key;
        idx_out[slot] = src;
    }
}

// Leaves the rays' indices in key order in the buffer it returns (a.idx[0] or a.idx[1]).
const uint32_t* launch_ray_sort(const RaySortArgs& a, int grid, hipStream_t stream)
{
    const uint32_t key_bits = (a.dir_bits ? 2u * a.dir_bits : 3u) + 3u * a.cell_bits;
    const int passes        = (int)((key_bits + 7u) / 8u);
    const dim3 g((unsigned)grid), b(256);
    for (int p = 0; p < passes; ++p) {
        const uint32_t shift = 8u * (uint32_t)p;
        const uint32_t* kin  = a.keys[p & 1];
        const uint32_t* iin  = a.idx[p & 1];
        uint32_t* kout       = a.keys[(p & 1) ^ 1];
        uint32_t* iout       = a.idx[(p & 1) ^ 1];
        const bool first = p == 0, last = p + 1 == passes;
        if (first)
            hipLaunchKernelGGL(k_rs_count<true>, g, b, 0, stream, a, kin, shift);
        else
            hipLaunchKernelGGL(k_rs_count<false>, g, b, 0, stream, a, kin, shift);
        hipLaunchKernelGGL(k_rs_prefix, dim3(256), b, 0, stream, a, (uint32_t)grid);
        if (first && last)
            hipLaunchKernelGGL((k_rs_scatter<true, true>), g, b, 0, stream, a, kin, iin, kout, iout, shift);
        else if (first)
            hipLaunchKernelGGL((k_rs_scatter<true, false>), g, b, 0, stream, a, kin, iin, kout, iout, shift);
        else if (last)
            hipLaunchKernelGGL((k_rs_scatter<false, true>), g, b, 0, stream, a, kin, iin, kout, iout, shift);
        else
            hipLaunchKernelGGL((k_rs_scatter<false, false>), g, b, 0, stream, a, kin, iin, kout, iout, shift);
    }
    return a.idx[passes & 1];
}

} // namespace igdev
