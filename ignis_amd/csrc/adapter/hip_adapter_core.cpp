// hip_adapter_core.cpp — see hip_adapter_core.h. Plain C++17 over include/igd_device.h and include/igh_host.h.
#include "hip_adapter_core.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace igadapter {

Core::Core(int gpu_index, bool acquire_stats, bool debug_trace, bool is_interactive)
{
    igd_setup setup{};
    setup.gpu_index      = gpu_index;
    setup.acquire_stats  = acquire_stats ? 1 : 0;
    setup.debug_trace    = debug_trace ? 1 : 0;
    setup.is_interactive = is_interactive ? 1 : 0;
    // "Normals" / "Albedo" are asked for by name only when the runtime's denoiser is on; keeping them costs two film buffers and
    // one extra camera-ray traversal at iteration 0
    setup.info_aovs = 1;
    // IRenderDevice::render returns when the iteration is done (SURVEY 8b Threading; the runtime's Statistics timers sit around
    // the call), so the adapter blocks like the reference's devices do. A frontend that only reads results at the end can opt
    // into deferral — consecutive iterations batched into one wavefront (igd_device.h), +25 % on the headline workload — with
    // IG_HIP_DEFERRED_RENDER=1. Interactive setups are never deferred by the device anyway.
    const char* deferred  = std::getenv("IG_HIP_DEFERRED_RENDER");
    setup.blocking_render = (deferred && *deferred && *deferred != '0') ? 0 : 1;
    mDev                  = igd_create(&setup);
    if (!mDev)
        mError = igd_last_error();
}

Core::~Core()
{
    if (mDev)
        igd_destroy(mDev);
    igh_free(mScene);
}

bool Core::fail(const std::string& what)
{
    mError = what;
    return false;
}

bool Core::setSceneFile(const std::string& path)
{
    igh_free(mScene);
    mScene = igh_load_file(path.c_str(), nullptr);
    return mScene ? true : fail(igh_last_error());
}

bool Core::setSceneString(const std::string& json, const std::string& base_dir)
{
    igh_free(mScene);
    mScene = igh_load_string(json.c_str(), base_dir.c_str(), nullptr);
    return mScene ? true : fail(igh_last_error());
}

std::string checkDatabase(const DatabaseView& db, const igd_scene& low)
{
    if (!db.entities.data || !db.shape_data.data || !db.primbvh.data || !db.scene_nodes.data || !db.scene_leaves.data)
        return "a table is missing";
    if (db.scene_nodes.size % sizeof(ig_node8) != 0 || db.scene_leaves.size % sizeof(ig_entity_leaf1) != 0)
        return "the scene BVH is not in the <8, 4> layout (Node8 = 256 B, EntityLeaf1 = 96 B); a GPU target builds BVH2 tables";
    if (db.entities.size != (size_t)low.entity_count * IG_ENTITY_FLOATS * sizeof(float))
        return "entity count differs from the lowered scene description";
    if (db.shape_lookups.size != (size_t)low.shape_count * sizeof(ig_lookup_entry))
        return "shape count differs from the lowered scene description";
    if (db.scene_leaves.size / sizeof(ig_entity_leaf1) != low.scene_leaf_count)
        return "scene BVH leaf count differs from the lowered scene description";
    // Entity and material ids follow std::unordered_map order in the reference and declaration order in
    // the host library: the PODs are indexed by material id, so the runtime's tables are usable only when both orders agree.
    // Per entity: same shape id and material id, same transform.
    if (std::memcmp(db.entities.data, low.entities, db.entities.size) != 0)
        return "entity records (transforms / shape ids / material ids) differ: the runtime numbered entities differently";
    if (db.entity_per_material) {
        if (db.material_count > low.material_count)
            return "material count differs from the lowered scene description";
        std::vector<int32_t> count(low.material_count, 0);
        for (uint32_t e = 0; e < low.entity_count; ++e) {
            int32_t m;
            std::memcpy(&m, low.entities + (size_t)e * IG_ENTITY_FLOATS + 34, 4);
            if (m >= 0 && (uint32_t)m < low.material_count)
                ++count[m];
        }
        for (size_t m = 0; m < db.material_count; ++m)
            if (db.entity_per_material[m] != count[m])
                return "entity_per_material differs from the lowered scene description";
    }
    // the prim BVH offsets of the leaves must point into the runtime's fix table
    const auto* leaves = reinterpret_cast<const ig_entity_leaf1*>(db.scene_leaves.data);
    for (size_t i = 0; i < db.scene_leaves.size / sizeof(ig_entity_leaf1); ++i) {
        const uint64_t off = (((uint64_t)(uint32_t)leaves[i].user[1] << 32) | (uint64_t)(uint32_t)leaves[i].user[0]) * 4;
        if (off + 16 > db.primbvh.size)
            return "a scene BVH leaf points outside the prim BVH table";
    }
    return {};
}

bool Core::assignScene(const DatabaseView* db)
{
    if (!mDev)
        return fail("no device");
    if (!mScene)
        return fail("no scene description: call setSceneFile / setSceneString before assignScene (INTEGRATION.md, Runtime hook)");
    const igd_scene* low = igh_tables(mScene);
    igd_scene s          = *low;
    mUsedRuntimeTables   = false;
    if (db) {
        const std::string why = checkDatabase(*db, *low);
        if (why.empty()) {
            s.entities         = reinterpret_cast<const float*>(db->entities.data);
            s.shape_lookups    = reinterpret_cast<const ig_lookup_entry*>(db->shape_lookups.data);
            s.shape_data       = db->shape_data.data;
            s.shape_data_size  = db->shape_data.size;
            s.primbvh          = db->primbvh.data;
            s.primbvh_size     = db->primbvh.size;
            s.scene_nodes      = reinterpret_cast<const ig_node8*>(db->scene_nodes.data);
            s.scene_node_count = (uint32_t)(db->scene_nodes.size / sizeof(ig_node8));
            s.scene_leaves     = reinterpret_cast<const ig_entity_leaf1*>(db->scene_leaves.data);
            s.scene_leaf_count = (uint32_t)(db->scene_leaves.size / sizeof(ig_entity_leaf1));
            if (db->scene_radius > 0)
                s.scene_radius = db->scene_radius;
            mUsedRuntimeTables = true;
        } else {
            mError = "runtime tables not used (" + why + "); using the host library's";
        }
    }
    if (igd_assign_scene(mDev, &s) != IGD_OK)
        return fail(igd_last_error());
    return true;
}

void Core::forwardInt(const char* name, int32_t v) { igd_set_parameter_i32(mDev, name, v); }
void Core::forwardFloat(const char* name, float v) { igd_set_parameter_f32(mDev, name, v); }
void Core::forwardVector(const char* name, float x, float y, float z)
{
    const float v[3] = { x, y, z };
    igd_set_parameter_vec3(mDev, name, v);
}

bool Core::render(const PlainRay* rays, size_t spi, size_t width, size_t height, size_t iteration, size_t frame, size_t user_seed)
{
    igd_render_settings s{};
    if (rays) { // Runtime::trace: width = #rays, height = 1 (Runtime.cpp:389-446); directions normalised as Device.cpp:602-643 does
        mRayScratch.resize(width * 8);
        for (size_t i = 0; i < width; ++i) {
            const PlainRay& r = rays[i];
            const float len   = std::sqrt(r.direction[0] * r.direction[0] + r.direction[1] * r.direction[1] + r.direction[2] * r.direction[2]);
            const float inv   = len > 0 ? 1 / len : 0;
            float* o          = mRayScratch.data() + i * 8;
            o[0] = r.origin[0], o[1] = r.origin[1], o[2] = r.origin[2];
            o[3] = r.direction[0] * inv, o[4] = r.direction[1] * inv, o[5] = r.direction[2] * inv;
            o[6] = r.range[0], o[7] = r.range[1];
        }
        s.rays = mRayScratch.data();
    }
    s.spi = (int32_t)spi, s.width = (int32_t)width, s.height = (int32_t)height;
    s.iteration = (int32_t)iteration, s.frame = (int32_t)frame, s.user_seed = (int32_t)user_seed;
    s.row_offset = 0, s.row_stride = 1;
    if (igd_render(mDev, &s) != IGD_OK)
        return fail(igd_last_error());
    return true;
}

void Core::resize(size_t w, size_t h) { igd_resize(mDev, (int32_t)w, (int32_t)h); }
void Core::releaseAll() { igd_release_all(mDev); }
size_t Core::framebufferWidth() const { return (size_t)igd_framebuffer_width(mDev); }
size_t Core::framebufferHeight() const { return (size_t)igd_framebuffer_height(mDev); }
float* Core::framebufferForHost(const std::string& name, bool sync) { return const_cast<float*>(igd_framebuffer_host(mDev, name.c_str(), sync ? 1 : 0)); }
float* Core::framebufferForDevice(const std::string& name) { return igd_framebuffer_device(mDev, name.c_str()); }
void Core::clearFramebuffer(const std::string& name) { igd_clear_framebuffer(mDev, name.c_str()); }
void Core::clearAllFramebuffer()
{
    igd_clear_framebuffer(mDev, nullptr);
    for (const char* n : { "Normals", "Albedo" })
        igd_clear_framebuffer(mDev, n);
    if (igd_buffer_size(mDev, "Denoised")) // (exists once the runtime's denoiser has asked for it, extra/OIDN.cpp:106)
        igd_clear_framebuffer(mDev, "Denoised");
}
void Core::syncFramebufferHostToDevice(const std::string& name)
{
    if (const float* host = igd_framebuffer_host(mDev, name.c_str(), 0))
        igd_sync_framebuffer_to_device(mDev, name.c_str(), host);
}

size_t Core::bufferSizeInBytes(const std::string& name) { return (size_t)igd_buffer_size(mDev, name.c_str()); }
bool Core::copyBufferToHost(const std::string& name, void* dst, size_t max_bytes) { return igd_buffer_copy(mDev, name.c_str(), dst, max_bytes) == IGD_OK; }
void* Core::bufferForDevice(const std::string& name, size_t* size)
{
    uint64_t n = 0;
    void* p    = igd_buffer_ptr(mDev, name.c_str(), &n);
    if (size)
        *size = (size_t)n;
    return p;
}

void Core::drainStatistics(StatsSink& sink)
{
    igd_stats st{};
    if (igd_get_stats(mDev, &st) != IGD_OK) {
        mError = igd_last_error();
        return;
    }
    igd_reset_stats(mDev);
    // Quantity (Statistics.h:57-64): CameraRayCount = 0, ShadowRayCount = 1, BounceRayCount = 2
    sink.quantity(0, st.camera_rays);
    sink.quantity(1, st.shadow_rays);
    sink.quantity(2, st.bounce_rays);
    // ShaderType (Statistics.h:9-26): PrimaryTraversal = 1, SecondaryTraversal = 2, RayGeneration = 3, Hit = 4
    sink.shader(1, st.traverse_primary_launches, st.camera_rays + st.bounce_rays, st.ms_traverse_primary);
    sink.shader(2, st.traverse_secondary_launches, st.shadow_rays, st.ms_traverse_secondary);
    sink.shader(3, st.rounds ? 1 : 0, st.camera_rays, st.ms_generate);
    sink.shader(4, st.rounds, st.camera_rays + st.bounce_rays, st.ms_shade); // sort + hit + miss + compaction are one kernel here
}

} // namespace igadapter

// ---- C entry points for the unit tests (tests/test_adapter_core.py drive the class through ctypes)
extern "C" {
using namespace igadapter;

void* iga_create(int gpu_index, int acquire_stats, int interactive) { return new Core(gpu_index, acquire_stats != 0, false, interactive != 0); }
void iga_destroy(void* c) { delete static_cast<Core*>(c); }
int iga_ok(void* c) { return static_cast<Core*>(c)->ok() ? 1 : 0; }
const char* iga_error(void* c) { return static_cast<Core*>(c)->error().c_str(); }
int iga_set_scene_file(void* c, const char* path) { return static_cast<Core*>(c)->setSceneFile(path) ? 1 : 0; }
// tables: {entities, shape_lookups, shape_data, primbvh, scene_nodes, scene_leaves} as (pointer, size) pairs; NULL = no database
int iga_assign_scene(void* c, const void* const* ptrs, const uint64_t* sizes, const int32_t* entity_per_material, uint64_t material_count)
{
    if (!ptrs)
        return static_cast<Core*>(c)->assignScene(nullptr) ? 1 : 0;
    DatabaseView db;
    Bytes* f[6] = { &db.entities, &db.shape_lookups, &db.shape_data, &db.primbvh, &db.scene_nodes, &db.scene_leaves };
    for (int i = 0; i < 6; ++i)
        *f[i] = Bytes{ static_cast<const uint8_t*>(ptrs[i]), (size_t)sizes[i] };
    db.entity_per_material = entity_per_material;
    db.material_count      = (size_t)material_count;
    return static_cast<Core*>(c)->assignScene(&db) ? 1 : 0;
}
int iga_used_runtime_tables(void* c) { return static_cast<Core*>(c)->usedRuntimeTables() ? 1 : 0; }
// checkDatabase against the description loaded into `c` without touching a device
const char* iga_check_database(void* scene /* igh_scene* */, const void* const* ptrs, const uint64_t* sizes, const int32_t* entity_per_material, uint64_t material_count)
{
    static thread_local std::string msg;
    DatabaseView db;
    Bytes* f[6] = { &db.entities, &db.shape_lookups, &db.shape_data, &db.primbvh, &db.scene_nodes, &db.scene_leaves };
    for (int i = 0; i < 6; ++i)
        *f[i] = Bytes{ static_cast<const uint8_t*>(ptrs[i]), (size_t)sizes[i] };
    db.entity_per_material = entity_per_material;
    db.material_count      = (size_t)material_count;
    msg                    = checkDatabase(db, *igh_tables(static_cast<igh_scene*>(scene)));
    return msg.c_str();
}
void iga_forward_int(void* c, const char* n, int32_t v) { static_cast<Core*>(c)->forwardInt(n, v); }
void iga_forward_float(void* c, const char* n, float v) { static_cast<Core*>(c)->forwardFloat(n, v); }
void iga_forward_vector(void* c, const char* n, float x, float y, float z) { static_cast<Core*>(c)->forwardVector(n, x, y, z); }
int iga_render(void* c, const float* rays8 /* origin, direction, range per ray or NULL */, uint64_t spi, uint64_t w, uint64_t h, uint64_t it, uint64_t frame, uint64_t seed)
{
    return static_cast<Core*>(c)->render(reinterpret_cast<const PlainRay*>(rays8), spi, w, h, it, frame, seed) ? 1 : 0;
}
float* iga_framebuffer_host(void* c, const char* name) { return static_cast<Core*>(c)->framebufferForHost(name ? name : "", true); }
uint64_t iga_buffer_size(void* c, const char* name) { return static_cast<Core*>(c)->bufferSizeInBytes(name); }
int iga_copy_buffer(void* c, const char* name, void* dst, uint64_t max_bytes) { return static_cast<Core*>(c)->copyBufferToHost(name, dst, max_bytes) ? 1 : 0; }
// statistics drained into a flat array: [camera, shadow, bounce, then per shader type 1..4: launches, workload]
void iga_drain_statistics(void* c, uint64_t out[11])
{
    struct Flat final : StatsSink {
        uint64_t* o;
        void quantity(int q, uint64_t v) override { o[q] = v; }
        void shader(int t, uint64_t launches, uint64_t workload, double) override { o[3 + 2 * (t - 1)] = launches, o[4 + 2 * (t - 1)] = workload; }
    } sink;
    sink.o = out;
    static_cast<Core*>(c)->drainStatistics(sink);
}
}
