// HipRenderDevice.cpp — the reference-facing shell of the drop-in: IG::IRenderDevice / IG::IDeviceInterface for the HIP
// backend, so the reference runtime's DeviceManager (src/runtime/device/DeviceManager.cpp:65-258) can dlopen it as
// `ig_device_hip.so` and pick it for `--gpu-arch amd` (GPUArchitecture::AMD_HSA, src/runtime/device/Target.cpp:43-54).
//
// UNCOMPILED IN THIS REPOSITORY. It includes the reference's headers, which all reach Eigen through IG_Config.h; Eigen is
// not available here and no stand-in is written for it. Everything that could be written without those headers lives in
// hip_adapter_core.{h,cpp} (compiled by `make`, unit-tested): this file only converts types —
//   std::string / std::vector<uint8> tables of SceneDatabase  -> igadapter::DatabaseView (pointers + sizes)
//   ParameterSet maps, Eigen vectors                           -> forwardInt / forwardFloat / forwardVector
//   IG::Ray                                                    -> igadapter::PlainRay
//   igadapter::StatsSink                                       -> IG::Statistics::increase / beginShaderLaunch
// Build: add to src/device/CMakeLists.txt as INTEGRATION.md shows, linking libig_adapter_core, libig_device_hip, libig_host.
#include "device/IDeviceInterface.h" // reference: src/runtime/device/IDeviceInterface.h
#include "device/IRenderDevice.h"    // reference: src/runtime/device/IRenderDevice.h
#include "Logger.h"
#include "Statistics.h"
#include "table/SceneDatabase.h"

#include "hip_adapter_core.h"

#include <cstdlib>

namespace IG {

// The one addition the reference runtime needs: IRenderDevice has no channel for the scene DESCRIPTION —
// devices receive materials / lights / camera / technique as JIT-compiled Artic — so Runtime::loadFromFile / loadFromString
// hand it to devices that implement this interface (dynamic_cast) before assignScene().
class ISceneDescriptionSink {
public:
    virtual ~ISceneDescriptionSink()                                                       = default;
    virtual bool setSceneFile(const Path& path)                                            = 0;
    virtual bool setSceneString(const std::string& json, const Path& base_dir)             = 0;
};

class HipRenderDevice final : public IRenderDevice, public ISceneDescriptionSink {
public:
    explicit HipRenderDevice(const SetupSettings& s)
        : mSetup(s)
        , mCore((int)s.target.device(), s.AcquireStats, s.DebugTrace, s.IsInteractive)
    {
        if (!mCore.ok()) {
            IG_LOG(L_FATAL) << "ig_device_hip: " << mCore.error() << std::endl;
            std::abort(); // the reference aborts on unrecoverable device errors (Device.cpp:303-306)
        }
    }

    bool setSceneFile(const Path& path) override { return report(mCore.setSceneFile(path.generic_string())); }
    bool setSceneString(const std::string& json, const Path& base_dir) override { return report(mCore.setSceneString(json, base_dir.generic_string())); }

    void assignScene(const SceneSettings& settings) override
    {
        igadapter::DatabaseView view;
        const igadapter::DatabaseView* db = nullptr;
        if (settings.database) {
            const SceneDatabase& d = *settings.database;
            auto fix               = [&](const char* n) { const auto it = d.FixTables.find(n); return it == d.FixTables.end() ? igadapter::Bytes{} : igadapter::Bytes{ it->second.data().data(), it->second.data().size() }; };
            view.entities          = fix("entities");
            view.primbvh           = fix("trimesh_primbvh");
            if (const auto it = d.DynTables.find("shapes"); it != d.DynTables.end()) {
                view.shape_lookups = { reinterpret_cast<const uint8_t*>(it->second.lookups().data()), it->second.lookups().size() * sizeof(LookupEntry) };
                view.shape_data    = { it->second.data().data(), it->second.data().size() };
            }
            if (const auto it = d.SceneBVHs.find("trimesh"); it != d.SceneBVHs.end()) {
                view.scene_nodes  = { it->second.Nodes.data(), it->second.Nodes.size() };
                view.scene_leaves = { it->second.Leaves.data(), it->second.Leaves.size() };
            }
            if (settings.entity_per_material) {
                view.entity_per_material = settings.entity_per_material->data();
                view.material_count      = settings.entity_per_material->size();
            }
            view.scene_radius = d.SceneRadius;
            db                = &view;
        }
        if (report(mCore.assignScene(db)) && db && !mCore.usedRuntimeTables())
            IG_LOG(L_DEBUG) << "ig_device_hip: " << mCore.error() << std::endl;
    }

    void render(const TechniqueVariantShaderSet&, const RenderSettings& rs, ParameterSet* params) override
    {
        if (params) { // the global registry of this iteration (Runtime.cpp:366-387); unchanged values do not break the device's batch
            for (const auto& p : params->IntParameters)
                mCore.forwardInt(p.first.c_str(), p.second);
            for (const auto& p : params->FloatParameters)
                mCore.forwardFloat(p.first.c_str(), p.second);
            for (const auto& p : params->VectorParameters)
                mCore.forwardVector(p.first.c_str(), p.second.x(), p.second.y(), p.second.z());
        }
        std::vector<igadapter::PlainRay> rays;
        if (rs.rays) {
            rays.resize(rs.width);
            for (size_t i = 0; i < rs.width; ++i) {
                const Ray& r = rs.rays[i];
                rays[i]      = igadapter::PlainRay{ { r.Origin.x(), r.Origin.y(), r.Origin.z() }, { r.Direction.x(), r.Direction.y(), r.Direction.z() }, { r.Range.x(), r.Range.y() } };
            }
        }
        report(mCore.render(rs.rays ? rays.data() : nullptr, rs.spi, rs.width, rs.height, rs.iteration, rs.frame, rs.user_seed));
    }

    void resize(size_t w, size_t h) override { mCore.resize(w, h); }
    void releaseAll() override { mCore.releaseAll(); }

    Target target() const override { return mSetup.target; }
    size_t framebufferWidth() const override { return mCore.framebufferWidth(); }
    size_t framebufferHeight() const override { return mCore.framebufferHeight(); }
    bool isInteractive() const override { return mSetup.IsInteractive; }

    AOVAccessor getFramebufferForHost(const std::string& name, bool sync) override { return AOVAccessor{ mCore.framebufferForHost(name, sync) }; }
    AOVAccessor getFramebufferForDevice(const std::string& name, bool) override { return AOVAccessor{ mCore.framebufferForDevice(name) }; }
    void clearFramebuffer(const std::string& name) override { mCore.clearFramebuffer(name); }
    void clearAllFramebuffer() override { mCore.clearAllFramebuffer(); }
    void syncFramebufferHostToDevice(const std::string& name) override { mCore.syncFramebufferHostToDevice(name); }
    void syncAllFramebufferHostToDevice() override { mCore.syncFramebufferHostToDevice({}); }

    size_t getBufferSizeInBytes(const std::string& name) override { return mCore.bufferSizeInBytes(name); }
    bool copyBufferToHost(const std::string& name, void* dst, size_t max_bytes) override { return mCore.copyBufferToHost(name, dst, max_bytes); }
    BufferAccessor getBufferForDevice(const std::string& name) override
    {
        size_t n = 0;
        void* p  = mCore.bufferForDevice(name, &n);
        return BufferAccessor{ p, n };
    }

    const Statistics* getStatistics() override
    {
        // Statistics::dump (igcli --stats, Statistics.cpp:286-290) reads the three quantities and the per-shader launch counts /
        // workloads; its times come from its own host timers around begin / endShaderLaunch, so device times go to the log
        struct Sink final : igadapter::StatsSink {
            Statistics* s;
            void quantity(int q, uint64_t v) override { s->increase((Quantity)q, v); }
            void shader(int t, uint64_t launches, uint64_t workload, double ms) override
            {
                for (uint64_t i = 0; i < launches; ++i) {
                    s->beginShaderLaunch((ShaderType)t, (size_t)(workload / launches), 0);
                    s->endShaderLaunch((ShaderType)t, 0);
                }
                if (launches)
                    IG_LOG(L_DEBUG) << "ig_device_hip: shader type " << t << ": " << ms << " ms of device time in " << launches << " launches" << std::endl;
            }
        } sink;
        sink.s = &mStatistics;
        mCore.drainStatistics(sink);
        return &mStatistics;
    }

    // JIT entry points outside the hot path (OUT)
    void tonemap(uint32_t*, const TonemapSettings&) override { IG_LOG(L_ERROR) << "ig_device_hip: tonemap is not part of the HIP backend" << std::endl; }
    ImageInfoOutput imageinfo(const ImageInfoSettings&) override { return ImageInfoOutput{}; }
    void bake(const ShaderOutput<void*>&, const std::vector<std::string>*, float*) override {}
    void runPass(const ShaderOutput<void*>&) override {}

private:
    bool report(bool ok)
    {
        if (!ok)
            IG_LOG(L_ERROR) << "ig_device_hip: " << mCore.error() << std::endl;
        return ok;
    }
    SetupSettings mSetup;
    igadapter::Core mCore;
    Statistics mStatistics;
};

// ICompilerDevice: nothing is compiled at run time (kernels are AOT-built for gfx950); a non-null token
// keeps ScriptCompiler happy (src/runtime/device/ICompilerDevice.h).
class HipCompilerDevice final : public ICompilerDevice {
public:
    bool compile(const Settings&, const std::string&) const override { return true; }
    void* compileAndGet(const Settings&, const std::string&, const std::string&) const override
    {
        static int token;
        return &token;
    }
};

class HipDeviceInterface final : public IDeviceInterface {
public:
    Build::Version getVersion() const override { return Build::getVersion(); } // DeviceManager.cpp:180-194
    TargetArchitecture getArchitecture() const override { return GPUArchitecture::AMD_HSA; }
    IRenderDevice* createRenderDevice(const IRenderDevice::SetupSettings& s) const override { return new HipRenderDevice(s); }
    ICompilerDevice* createCompilerDevice() const override { return new HipCompilerDevice(); }
};
} // namespace IG

// src/device/Interface.cpp:70-76
extern "C" IG_EXPORT const IG::IDeviceInterface* ig_get_interface()
{
    static IG::HipDeviceInterface interface;
    return &interface;
}
