// hip_adapter_core.h — everything of the IRenderDevice adapter that does not need a reference header.
//
// The reference's plugin boundary (src/runtime/device/IRenderDevice.h:14-81) is a C++ ABI over IG_Config.h, i.e. over Eigen:
// none of those headers compiles in this repository (Eigen is absent). The adapter is therefore split:
//   * this file + hip_adapter_core.cpp: plain C++17 on top of the two C ABIs (include/igd_device.h, include/igh_host.h) —
//     scene hand-over from the runtime's SceneDatabase tables, registry forwarding, statistics mapping, named buffers.
//     Built by `make` into libig_adapter_core.so and unit-tested (tests/test_adapter_core.py).
//   * HipRenderDevice.cpp: the shell that derives from IG::IRenderDevice and converts reference types (std::string,
//     Eigen vectors, ParameterSet, SceneDatabase, Statistics) to the plain types below. It holds no logic of its own and is
//     compiled only inside a reference build tree; that part remains UNCOMPILED here.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "igd_device.h"
#include "igh_host.h"

namespace igadapter {

// A table of the runtime's SceneDatabase as bytes (FixTable::data(), DynTable::data() / lookups(), SceneBVH::Nodes / Leaves;
// src/runtime/table/{FixTable,DynTable,SceneDatabase}.h). Borrowed: IRenderDevice::SceneSettings says the database outlives
// the renders (IRenderDevice.h:23-28).
struct Bytes {
    const uint8_t* data = nullptr;
    size_t size         = 0;
};
struct DatabaseView {
    Bytes entities;        // FixTables["entities"]: 36 floats per entity (LoaderEntity.cpp:150-162)
    Bytes shape_lookups;   // DynTables["shapes"].lookups(): LookupEntry {u32 TypeID, u32 Flags, u64 Offset}
    Bytes shape_data;      // DynTables["shapes"].data()
    Bytes primbvh;         // FixTables["trimesh_primbvh"]
    Bytes scene_nodes;     // SceneBVHs["trimesh"].Nodes
    Bytes scene_leaves;    // SceneBVHs["trimesh"].Leaves
    const int32_t* entity_per_material = nullptr; // SceneSettings::entity_per_material
    size_t material_count              = 0;
    float scene_radius                 = 0;
};

// Ray of Runtime::trace (RuntimeStructs.h:39-43) without Eigen
struct PlainRay {
    float origin[3], direction[3], range[2];
};

// What Statistics (src/runtime/Statistics.h:57-64,66-152) can take from this device: the three quantities and, per shader
// type, launch count + workload. (Statistics measures times with its own host timers between begin / endShaderLaunch; a
// device-side elapsed time cannot be injected, so the HIP-event stage times are reported through the log instead.)
struct StatsSink {
    virtual ~StatsSink() = default;
    virtual void quantity(int quantity /* 0 camera, 1 shadow, 2 bounce (Statistics.h:57-64) */, uint64_t value) = 0;
    virtual void shader(int shader_type /* ShaderType (Statistics.h:9-26) */, uint64_t launches, uint64_t workload, double device_ms) = 0;
};

class Core {
public:
    Core(int gpu_index, bool acquire_stats, bool debug_trace, bool is_interactive);
    ~Core();
    Core(const Core&)            = delete;
    Core& operator=(const Core&) = delete;

    bool ok() const { return mDev != nullptr; }
    const std::string& error() const { return mError; }

    // Scene description the runtime loaded (one hook in Runtime::load*): materials, lights, camera and
    // technique exist in the reference only as generated Artic source, so the PODs are lowered from the description itself.
    bool setSceneFile(const std::string& path);
    bool setSceneString(const std::string& json, const std::string& base_dir);

    // IRenderDevice::assignScene: geometry from the runtime's own tables when they are in the <8, 4> layout this device
    // traverses (Node8 256 B / Tri4 208 B / EntityLeaf1 96 B: byte-compatible, include/ig_tables.h) and agree with the lowered
    // scene; otherwise (BVH2 tables of a GPU target, or a null database) the tables the host library built. Returns false on error.
    bool assignScene(const DatabaseView* db);
    bool usedRuntimeTables() const { return mUsedRuntimeTables; }

    // IRenderDevice::render. Registry values first (forwardInt / Float / Vector for every entry of the ParameterSet), then this.
    void forwardInt(const char* name, int32_t v);
    void forwardFloat(const char* name, float v);
    void forwardVector(const char* name, float x, float y, float z);
    bool render(const PlainRay* rays, size_t spi, size_t width, size_t height, size_t iteration, size_t frame, size_t user_seed);

    void resize(size_t w, size_t h);
    void releaseAll();
    size_t framebufferWidth() const;
    size_t framebufferHeight() const;
    float* framebufferForHost(const std::string& name, bool sync);
    float* framebufferForDevice(const std::string& name);
    void clearFramebuffer(const std::string& name);
    void clearAllFramebuffer();
    void syncFramebufferHostToDevice(const std::string& name);

    size_t bufferSizeInBytes(const std::string& name);
    bool copyBufferToHost(const std::string& name, void* dst, size_t max_bytes);
    void* bufferForDevice(const std::string& name, size_t* size);

    // IRenderDevice::getStatistics: what accumulated since the last call goes into the sink (which adds it to an
    // IG::Statistics), so repeated calls never count anything twice.
    void drainStatistics(StatsSink& sink);

private:
    bool fail(const std::string& what);
    igd_device* mDev  = nullptr;
    igh_scene* mScene = nullptr;
    bool mUsedRuntimeTables = false;
    std::string mError;
    std::vector<float> mRayScratch;
};

// Compares the geometry tables of a DatabaseView with a lowered igd_scene. Empty string = usable; otherwise why not.
std::string checkDatabase(const DatabaseView& db, const igd_scene& lowered);

} // namespace igadapter
