// photon.hip — the photon mapper's kernels (src/artic/technique/photonmapper.art): the two k_shade instantiations of its light and
// camera pass (ppm_core.h) and the construction of the query structure between them.
//
// ppm_handle_before_iteration_camera (:433-518) counts photons per cell of a 128^3 grid (Morton-linearised), scans the counts with
// 21 Hillis-Steele passes over two ping-pong buffers and scatters the photons with one atomic per photon — an unstable sort, so the
// order inside a cell depends on scheduling. Here: one 64-bit key (cell, light path index) per stored photon, ONE device radix sort
// (hipCUB), a gather of the photons into that order, counts by atomics (an integer result is order-independent) and one exclusive
// scan. The order inside a cell is the photons' index order: gathers sum in a fixed order and the image is reproducible.
#include <hipcub/hipcub.hpp>

#include "shade_kernel.h"

namespace igdev {

template __global__ void k_shade<true, false, true, false, 1>(const ShadeArgs);
template __global__ void k_shade<true, false, true, false, 2>(const ShadeArgs);

void launch_shade_ppm(const ShadeArgs& args, int grid_blocks, hipStream_t stream)
{
    if (args.ppm.pass == 1)
        hipLaunchKernelGGL((k_shade<true, false, true, false, 1>), dim3((unsigned)grid_blocks), dim3(kShadeThreads), 0, stream, args);
    else
        hipLaunchKernelGGL((k_shade<true, false, true, false, 2>), dim3((unsigned)grid_blocks), dim3(kShadeThreads), 0, stream, args);
}

// keys of the stored photons: (cell << 32) | index; a light path that left no photon (light == -1) gets the all-ones key and sorts last
__global__ void __launch_bounds__(256) k_photon_keys(const igp_photon* photons, uint32_t n, PpmArgs grid, unsigned long long* keys, uint32_t* cell_count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const igp_photon p = photons[i];
    if (p.light < 0) {
        keys[i] = ~0ull;
        return;
    }
    const uint32_t cell = (uint32_t)igp_grid_cell(p.pos, grid.bbox_min, grid.bbox_max);
    keys[i]             = ((unsigned long long)cell << 32) | i;
    atomicAdd(&cell_count[cell], 1u);
}

// (the light paths without a photon carry the all-ones key and sort behind every stored one: their slots are not gathered)
__global__ void __launch_bounds__(256) k_photon_gather(const igp_photon* photons, const unsigned long long* sorted_keys, uint32_t n, igp_photon* sorted)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || sorted_keys[i] == ~0ull)
        return;
    const uint32_t src = (uint32_t)(sorted_keys[i] & 0xFFFFFFFFull);
    const int4* s      = reinterpret_cast<const int4*>(photons) + 2 * (size_t)src;
    int4* d            = reinterpret_cast<int4*>(sorted) + 2 * (size_t)i;
    d[0] = s[0], d[1] = s[1];
}

// scratch sizes of the two library calls for n photons
size_t photon_grid_temp_bytes(uint32_t n)
{
    size_t sort_bytes = 0, scan_bytes = 0;
    if (hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)n) != hipSuccess
        || hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, IGP_GRID_CELLS + 1) != hipSuccess)
        return 0;
    return sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
}

// photons[0 .. n) (slot = light path index) -> sorted[0 .. *valid) in (cell, index) order, cell_offset[0 .. IGP_GRID_CELLS]
hipError_t build_photon_grid(const igp_photon* photons, uint32_t n, const PpmArgs& grid, igp_photon* sorted, uint32_t* cell_count /* IGP_GRID_CELLS + 1 */,
                       uint32_t* cell_offset /* IGP_GRID_CELLS + 1 */, unsigned long long* keys /* 2 n */, uint32_t* valid, void* temp, size_t temp_bytes, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(cell_count, 0, (size_t)(IGP_GRID_CELLS + 1) * sizeof(uint32_t), stream);
    if (e != hipSuccess)
        return e;
    const unsigned blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_photon_keys, dim3(blocks ? blocks : 1u), dim3(256), 0, stream, photons, n, grid, keys, cell_count);
    size_t bytes = temp_bytes;
    // (21 bits of cell over 32 bits of index: the all-ones key of an empty slot still sorts behind every photon of the last cell)
    static_assert(IGP_GRID_CELLS == (1 << 21), "key layout");
    e = hipcub::DeviceRadixSort::SortKeys(temp, bytes, keys, keys + n, (int)n, 0, 53, stream);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(k_photon_gather, dim3(blocks ? blocks : 1u), dim3(256), 0, stream, photons, keys + n, n, sorted);
    bytes = temp_bytes;
    e     = hipcub::DeviceScan::ExclusiveSum(temp, bytes, cell_count, cell_offset, IGP_GRID_CELLS + 1, stream);
    if (e != hipSuccess)
        return e;
    // the number of stored photons is the scan's total (no counter of its own: it was one same-address atomic per photon)
    return hipMemcpyAsync(valid, cell_offset + IGP_GRID_CELLS, sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
}

// ---------------------------------------------------------------- the light tracer's connections

// K7 + K8 of the reference for the light tracer (gpu_sort_secondary: the shadow rays partitioned by verdict, and gpu_advanced_shadow:
// on_shadow_miss per unoccluded ray, mapping_gpu.art:293-333,504-614; technique/lighttracer.art:116-120 splats the connection into the
// film). A connection lands in the accumulator slot of ANOTHER path's pixel, so many rays of a round add into one slot; the reference
// (and rounds 1 - 4 here) do that with float atomics, in whatever order the hardware serves them. Here: the any-hit launch only
// records its verdicts, the unoccluded rays get the key (slot, light path id) — unique within a round: a path makes one connection
// per bounce — one radix sort puts them in that order (the occluded ones behind), and one thread per slot adds its run in order:
// the film of a given seed is the same bits run after run.
__global__ void __launch_bounds__(256) k_lt_keys(const float4* __restrict__ col, const float4* __restrict__ verdict, const uint32_t* __restrict__ path_id, const uint32_t* count,
                                                  uint32_t bound, unsigned long long* keys, uint32_t* vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bound)
        return;
    unsigned long long key = ~0ull;
    if (i < *count && (int)igm_bits(verdict[i].y) < 0) // (prim id of the any-hit launch: negative = nothing in the way)
        key = ((unsigned long long)igm_bits(col[i].w) << 32) | path_id[i];
    keys[i] = key;
    vals[i] = i;
}

__global__ void __launch_bounds__(256) k_lt_splat(const float4* __restrict__ col, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t bound,
                                                   float4* accum, int64_t id_base, float inv_spi)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= bound)
        return;
    const unsigned long long key = keys[j];
    if (key == ~0ull)
        return;
    const uint32_t slot = (uint32_t)(key >> 32);
    if (j > 0 && (uint32_t)(keys[j - 1] >> 32) == slot)
        return; // not the first of its slot's run
    float4* dst = accum + ((int64_t)(int32_t)slot - id_base);
    float4 v    = *dst;
    for (uint32_t k = j; k < bound && (uint32_t)(keys[k] >> 32) == slot && keys[k] != ~0ull; ++k) {
        const float4 c = col[vals[k]];
        v.x += c.x * inv_spi;
        v.y += c.y * inv_spi;
        v.z += c.z * inv_spi;
    }
    *dst = v;
}

size_t lt_splat_temp_bytes(uint32_t bound)
{
    size_t bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)bound) != hipSuccess)
        return 0;
    return bytes;
}

// col / verdict / path_id: the round's shadow rays (count on the device, at most `bound`); keys, vals: 2 x bound entries each
hipError_t launch_lt_splat(const float4* col, const float4* verdict, const uint32_t* path_id, const uint32_t* count, uint32_t bound, unsigned long long* keys, uint32_t* vals,
                           void* temp, size_t temp_bytes, float4* accum, int64_t id_base, float inv_spi, hipStream_t stream)
{
    if (bound == 0)
        return hipSuccess;
    const unsigned blocks = (bound + 255u) / 256u;
    hipLaunchKernelGGL(k_lt_keys, dim3(blocks), dim3(256), 0, stream, col, verdict, path_id, count, bound, keys, vals);
    size_t bytes       = temp_bytes;
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, bytes, keys, keys + bound, vals, vals + bound, (int)bound, 0, 64, stream);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(k_lt_splat, dim3(blocks), dim3(256), 0, stream, col, keys + bound, vals + bound, bound, accum, id_base, inv_spi);
    return hipGetLastError();
}

} // namespace igdev
