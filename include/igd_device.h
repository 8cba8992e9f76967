/* igd_device.h — C ABI of the MI355X render device (libig_device_hip.so).
 *
 * The reference's device boundary is the C++ plugin interface
 *   IDeviceInterface  src/runtime/device/IDeviceInterface.h:9-17
 *   IRenderDevice     src/runtime/device/IRenderDevice.h:14-81
 * loaded by DeviceManager through `ig_get_interface` (src/device/Interface.cpp:70-76).
 * Its signatures carry std::string / std::vector / Eigen types, so a plugin has
 * to be built with the runtime's own toolchain; this library exports the same
 * operations as a plain C ABI (pointers + sizes), and the ~200-line C++ adapter
 * in ignis_amd/csrc/adapter/ maps IRenderDevice onto it.
 * Each entry point names the reference method it replaces.
 *
 * Error convention: the reference logs and carries on (or std::abort()s on OOM,
 * src/device/Device.cpp:303-306). Here every call returns IGD_OK or a negative
 * code and igd_last_error() holds the message; nothing falls back to the CPU.
 */
#ifndef IGD_DEVICE_H
#define IGD_DEVICE_H

#include "ig_tables.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): igd_stats grew at its end (ms_ray_sort, stream_bytes) and igd_comm_available was added; a caller built against version 1
 * would hand igd_get_stats a smaller struct — what the reference's DeviceManager compares before it uses a device (DeviceManager.cpp:180-194). */
#define IGD_ABI_VERSION 2u

enum igd_status {
    IGD_OK                = 0,
    IGD_ERR_INVALID_ARG   = -1,
    IGD_ERR_NO_DEVICE     = -2, /* no HIP device / wrong architecture */
    IGD_ERR_OUT_OF_MEMORY = -3,
    IGD_ERR_NO_SCENE      = -4,
    IGD_ERR_UNSUPPORTED   = -5, /* scene feature the HIP backend cannot lower */
    IGD_ERR_DEVICE        = -6, /* HIP runtime error, traversal stack overflow, ... */
};

typedef struct igd_device igd_device;

/* IRenderDevice::SetupSettings (IRenderDevice.h:16-21) + GPU selection
 * (Target::device(), src/runtime/device/Target.h). */
typedef struct igd_setup {
    int32_t gpu_index;      /* HIP device ordinal */
    int32_t acquire_stats;  /* 0 off, 1 per-stage HIP-event timers, 2 timers + traversal work counters */
    int32_t debug_trace;
    int32_t is_interactive;
    uint64_t stream_capacity; /* rays in flight, allocated as given at the first render; 0 = grow with the largest request up to
                               * one batch of iterations (2^29 rays, or what free device memory allows); larger requests run in chunks (reference: 1 048 576,
                               * mapping_gpu.art:1119) */
    int32_t info_aovs;        /* != 0: the "Normals" and "Albedo" AOVs of the info-buffer wrapper the runtime adds for the denoiser
                               * (InfoBufferTechnique.cpp:6-18, technique/internal/infobuffer.art): first hits of the camera rays of
                               * iteration 0; read them through the framebuffer accessors by name */
    int32_t blocking_render;  /* != 0: igd_render executes its iteration and returns when it is complete, like the reference's
                               * IRenderDevice::render (SURVEY.md 8b "Threading"), errors included. 0 (default): igd_render records
                               * and defers, see igd_render below. is_interactive implies immediate execution (not completion). */
} igd_setup;

/* IRenderDevice::RenderSettings (IRenderDevice.h:30-40). `rays` != NULL selects the
 * list emitter of Runtime::trace (src/runtime/Runtime.cpp:389-446): width = #rays, height = 1.
 * Tile sharding (new): this device renders film rows
 * row_offset, row_offset + row_stride, ...; (0, 1) = the whole film. */
typedef struct igd_render_settings {
    const float* rays; /* host pointer, 8 floats per ray: org, dir, tmin, tmax; or NULL */
    int32_t spi;
    int32_t width, height;
    int32_t iteration, frame, user_seed;
    int32_t row_offset, row_stride;
    /* > 1: this call renders the iterations iteration, iteration + 1, ... as one wavefront (ray id = ((it * pixels + pixel)
     * * spi + sample)). The result is bit-identical to that many single-iteration calls; the point is efficiency when one
     * iteration is too small to fill the GPU (small films, row-sharded films). 0 or 1: one iteration, as the reference. */
    int32_t iterations;
} igd_render_settings;

/* Statistics (src/runtime/Statistics.h:57-64 quantities + ShaderType timers). */
typedef struct igd_stats {
    uint64_t camera_rays, bounce_rays, shadow_rays; /* shadow_rays counts real shadow rays */
    uint64_t unoccluded;                             /* shadow rays that reached the light */
    /* traversal work (inner nodes fetched, triangles tested, entity leaves tested), split by kernel;
     * only counted with acquire_stats >= 2 */
    uint64_t nodes_primary, tris_primary, leaves_primary;
    uint64_t nodes_secondary, tris_secondary, leaves_secondary;
    uint64_t traverse_primary_launches, traverse_secondary_launches;
    double ms_generate, ms_traverse_primary, ms_shade, ms_traverse_secondary, ms_resolve; /* HIP-event time, acquire_stats */
    double ms_total;   /* wall time inside igd_render, host clock */
    uint32_t rounds;   /* wavefront bounce rounds executed */
    uint32_t pad;
    uint64_t tail_rays; /* paths finished by the single-launch tail kernel instead of more rounds */
    double ms_tail;
    /* k_traverse executes its three sections (0 entity leaf, 1 inner node, 2 triangle packet) for whole waves under predicates:
     * section_passes counts wave-level executions, section_lanes the lanes that had work in them, [0..2] closest-hit launches,
     * [3..5] any-hit launches; lanes / (64 * passes) is the useful share of the issued section instructions (acquire_stats >= 2) */
    uint64_t section_passes[6], section_lanes[6];
    double ms_ray_sort; /* ordering bounce / shadow rays in space in front of their traversal launches (IGD_RAY_SORT=1) */
    /* Bytes per ray the wavefront kernels of the assigned scene and camera move BY CONSTRUCTION (the columns each stream kind carries,
     * csrc/device/kernels.h kStream*; what a roofline's byte model multiplies with the ray counts above):
     * [0] a camera ray read by the closest-hit launch (16: a one-point camera stores directions only; 48 otherwise)
     * [1] a bounce ray read by the closest-hit launch (32: origin and direction; its flags and tmax are uniform)
     * [2] the hit written per ray (16 packed, else 20)
     * [3] what k_shade reads per camera hit, [4] per bounce hit (the ray's columns + the hit; a skipped miss reads the hit only)
     * [5] a bounce ray written by k_shade (64), [6] a shadow ray written by k_shade and read by the any-hit launch (48)
     * [7] the accumulator traffic of an unoccluded shadow ray (16 read with the ray + 16 written) */
    uint32_t stream_bytes[8];
} igd_stats;

/* IDeviceInterface::getVersion (IDeviceInterface.h:11) */
uint32_t igd_get_abi_version(void);

/* Number of usable gfx950 devices; <= 0 means none (igd_last_error says why). */
int32_t igd_device_count(void);

/* IDeviceInterface::createRenderDevice (IDeviceInterface.h:14) / Device::Device (Device.cpp:1621-1640) */
igd_device* igd_create(const igd_setup* setup);
void igd_destroy(igd_device* dev);

/* IRenderDevice::assignScene (IRenderDevice.h:43, Device.cpp:1642-1670): uploads all tables once.
 * The scene is copied to HBM; the host pointers need not outlive this call. */
int32_t igd_assign_scene(igd_device* dev, const igd_scene* scene);

/* IRenderDevice::render (IRenderDevice.h:44, Device.cpp:1672-1682).
 * CONTRACT (differs from the reference unless igd_setup.blocking_render is set): the call validates its arguments — those
 * errors are returned here — and RECORDS the iteration; it may return before anything has run. Consecutive iterations of the
 * same film / spi / seed / sharding are executed together as one wavefront (see igd_synchronize below for when). What a caller
 * can observe is unchanged: every accessor of results (framebuffer, statistics, buffers) first executes and drains what is
 * pending, and the image is bit-identical to executing every call on its own. What a caller must NOT assume: that the time
 * spent inside igd_render is the iteration's render time, or that an error of the execution (device fault, traversal stack
 * overflow) is returned by the igd_render that caused it — it is returned by the next call that drains (any accessor,
 * igd_synchronize). With igd_setup.blocking_render != 0 the call executes and completes the iteration before it returns. */
int32_t igd_render(igd_device* dev, const igd_render_settings* settings);

/* IRenderDevice::resize (IRenderDevice.h:45): reallocates and clears the framebuffer. */
int32_t igd_resize(igd_device* dev, int32_t width, int32_t height);

/* IRenderDevice::releaseAll (IRenderDevice.h:47) */
int32_t igd_release_all(igd_device* dev);

int32_t igd_framebuffer_width(const igd_device* dev);  /* IRenderDevice::framebufferWidth */
int32_t igd_framebuffer_height(const igd_device* dev); /* IRenderDevice::framebufferHeight */

/* IRenderDevice::getFramebufferForHost(name, sync) (IRenderDevice.h:53, Device.cpp:1385-1417):
 * float[height][width][3], device -> host copy if dirty; pointer owned by the device, valid
 * until resize/destroy. name NULL or "" = colour buffer; "Normals" / "Albedo" with igd_setup.info_aovs. Returns NULL for
 * unknown AOVs.
 * Denoiser hook: with igd_setup.info_aovs the name "Denoised" addresses one more film-sized buffer the device never writes
 * (allocated zeroed at its first access). The runtime's denoiser reads the colour, "Normals" and "Albedo" buffers, writes its
 * result there and, on the host route, calls igd_sync_framebuffer_to_device("Denoised") -- exactly the four accessor calls of
 * extra/OIDN.cpp:103-106,123 (host) and :132-135 (device, OIDN's HIP device type shares the pointers), so OIDN.cpp needs no change. */
const float* igd_framebuffer_host(igd_device* dev, const char* name, int32_t sync);

/* IRenderDevice::getFramebufferForDevice (IRenderDevice.h:54): device pointer (HBM). */
float* igd_framebuffer_device(igd_device* dev, const char* name);

/* IRenderDevice::clearFramebuffer / clearAllFramebuffer (IRenderDevice.h:55-56) */
int32_t igd_clear_framebuffer(igd_device* dev, const char* name);

/* IRenderDevice::syncFramebufferHostToDevice (IRenderDevice.h:58): uploads `data`
 * (float[height][width][3]) into the device framebuffer (checkpoint resume). */
int32_t igd_sync_framebuffer_to_device(igd_device* dev, const char* name, const float* data);

/* IRenderDevice::getBufferSizeInBytes / copyBufferToHost / getBufferForDevice (IRenderDevice.h:62-64): the device-resident
 * tables by the names the reference gives them — "entities", "shapes", "trimesh_primbvh" (SceneDatabase tables,
 * src/runtime/table/SceneDatabase.h), "scene_bvh_nodes", "scene_bvh_leaves" — plus "materials", "lights" (ig_tables.h PODs) and
 * the film buffers "Color" / "Normals" / "Albedo". Unknown names: size 0, copy returns IGD_ERR_INVALID_ARG, ptr NULL (the
 * reference logs and returns an empty accessor, Device.cpp:1391-1395). The accessors drain pending work first. */
uint64_t igd_buffer_size(igd_device* dev, const char* name);
int32_t igd_buffer_copy(igd_device* dev, const char* name, void* dst, uint64_t max_bytes);
void* igd_buffer_ptr(igd_device* dev, const char* name, uint64_t* size_in_bytes);

/* IRenderDevice::getStatistics (IRenderDevice.h:66): cumulative since igd_reset_stats. */
int32_t igd_get_stats(igd_device* dev, igd_stats* out);
int32_t igd_reset_stats(igd_device* dev);

/* Stage dispatch of the traversal kernels on a ray list — what the reference reaches through
 * ignis_handle_traverse_primary / _secondary (src/device/Device.cpp:1020-1061,2044-2054) with the
 * list emitter (src/artic/driver/emitter.art:18-30). rays: host, 8 floats per ray (org, dir,
 * tmin, tmax). any_hit = 0: closest hit, outputs (ent_id, prim_id, t, u, v) as the primary
 * stream's hit columns (src/artic/driver/streams.art:16-20); any_hit = 1: prim_id >= 0 marks
 * an occluded ray. Output pointers are host arrays of `count` entries and may be NULL.
 * `repeat` >= 1 re-runs the kernel on the resident rays (timing); kernel_ms (may be NULL)
 * receives the average HIP-event duration of one launch. */
int32_t igd_traverse(igd_device* dev, int64_t count, const float* rays, uint32_t ray_flags, int32_t any_hit,
                     int32_t* ent_id, int32_t* prim_id, float* t, float* u, float* v,
                     int32_t repeat, double* kernel_ms);

/* The registry parameters IRenderDevice::render receives with every call (ParameterSet*, IRenderDevice.h:53;
 * filled by Runtime::setParameter / setCameraOrientation, Runtime.cpp:696-741) that this path reads:
 *   vec3 "__camera_eye" / "__camera_dir" / "__camera_up"   (PerspectiveCamera.cpp:69-76, camera/perspective.art)
 *   f32  "__camera_scale"                                  (OrthogonalCamera.cpp:28,39)
 *   i32  "__tech_max_depth" / "__tech_min_depth", f32 "__tech_clamp"   (PathTechnique.cpp:38-40)
 * They take effect from the next igd_render. Other names are accepted and ignored, as the reference's registry
 * stores parameters no shader reads. */
int32_t igd_set_parameter_i32(igd_device* dev, const char* name, int32_t value);
int32_t igd_set_parameter_f32(igd_device* dev, const char* name, float value);
int32_t igd_set_parameter_vec3(igd_device* dev, const char* name, const float value[3]);

/* igd_render validates its arguments and records the request; consecutive iterations of the same film, spi, seed and
 * sharding are executed together as one wavefront of up to 2^29 camera rays (the result is bit-identical to executing each
 * call on its own; IGD_BATCH_RAYS=0 in the environment or igd_setup.is_interactive make every call execute immediately).
 * The long-path tail and the framebuffer resolve of a wavefront run on other HIP streams under the next one (the
 * reference's render() is followed by getFramebufferForHost(), which is where it syncs, Device.cpp:1385-1425). Every
 * accessor that reads results (framebuffer_host/_device, get_stats, clear, resize, assign_scene, traverse), a parameter
 * change and igd_synchronize execute and drain what is pending; igd_synchronize does only that, and is where an error of
 * deferred or overlapped work (e.g. a traversal stack overflow) is reported if no other call has surfaced it yet.
 * IGD_ASYNC_TAIL=0 keeps the tail on the critical path. */
int32_t igd_synchronize(igd_device* dev);

/* ---- the one exchange step of a tile-sharded render (BASELINE.json north_star: "tile-sharded across GPUs with an RCCL gather over
 * xGMI only for final accumulation"). One process per GPU; rank r renders film rows r, r + N, ... (igd_render_settings.row_offset /
 * row_stride) and igd_comm_gather_rows moves each rank's rows into rank dst's framebuffer: ceil(H / N) x W x 12 bytes per rank,
 * grouped ncclSend / ncclRecv on the device's render stream, no arithmetic on the way. The reference has no counterpart (its
 * devices render whole films, src/runtime/device/IRenderDevice.h:44; Runtime::getFramebufferForHost, Runtime.cpp:438-455, is where
 * the gathered film is read). librccl.so is opened at run time by the device library itself (no torch, no MPI); the launcher hands
 * every rank the id rank 0 obtained (ignis_amd/comm.py). */
#define IGD_COMM_ID_BYTES 128
/* 1: librccl.so is loaded with every entry point this library calls; 0: not (igd_last_error says why). What a rank reports in the
 * launcher's bring-up vote before any rank enters ncclCommInitRank (ignis_amd/comm.py agree): a collective a peer cannot join never starts. */
int32_t igd_comm_available(void);
int32_t igd_comm_unique_id(uint8_t id[IGD_COMM_ID_BYTES]);                                                    /* ncclGetUniqueId */
int32_t igd_comm_init(igd_device* dev, const uint8_t id[IGD_COMM_ID_BYTES], int32_t rank, int32_t world_size); /* ncclCommInitRank on the device's GPU */
int32_t igd_comm_world_size(igd_device* dev);                                                                 /* ncclCommCount; 0: no communicator */
int32_t igd_comm_gather_rows(igd_device* dev, int32_t dst_rank);
/* values[i] <- sum (op 0) or max (op 2) over the ranks: ray statistics, the slowest rank's clock; a barrier as well */
int32_t igd_comm_allreduce_f64(igd_device* dev, double* values, int32_t count, int32_t op);
int32_t igd_comm_destroy(igd_device* dev);

/* Bytes of one inner BVH node as the traversal kernels of the assigned scene fetch it: 256 (the reference's Node8,
 * src/artic/traversal/bvh.art:85-89) or 128 (the same node on its 8-bit grid, when the scene's builder quantised the boxes and
 * the device packed them without loss; IGD_NODE_FORMAT=full keeps Node8). What a roofline prices a node visit at. 0: no scene. */
int32_t igd_node_bytes(const igd_device* dev);

/* Thread-local message of the last failed igd_* call ("" if none). */
const char* igd_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* IGD_DEVICE_H */
