// HipRenderDevice.cpp — the C++ side of the drop-in: IG::IRenderDevice / IG::IDeviceInterface
// implemented on top of the C ABI in include/igd_device.h, so the reference runtime's DeviceManager
// (src/runtime/device/DeviceManager.cpp:65-258) can dlopen it as `ig_device_hip.so` and pick it for
// `--gpu-arch amd` (GPUArchitecture::AMD_HSA, src/runtime/device/Target.cpp:43-54).
//
// This file is compiled ONLY inside a reference build tree (it needs the reference's headers, hence
// the same Eigen / STL as libig_runtime): add it to src/device/CMakeLists.txt as shown in
// INTEGRATION.md. It is not part of `make` in this repo (none of those headers exist here) and holds
// no rendering logic: every method is one igd_* call.
//
// What cannot cross this boundary as-is: the reference hands materials / lights / camera / technique
// to a device as JIT-compiled Artic (TechniqueVariantShaderSet of void* entry points). This backend
// needs them as PODs (include/ig_tables.h), so the adapter asks the companion host library
// (include/igh_host.h) to lower the scene FILE the runtime was given; the SceneDatabase tables the
// runtime built (Node8/Tri4 for a vector-width-8 CPU target) are byte-compatible and could be passed
// through instead once the loader exposes them for GPU targets.
#include "device/IDeviceInterface.h" // reference: src/runtime/device/IDeviceInterface.h
#include "device/IRenderDevice.h"    // reference: src/runtime/device/IRenderDevice.h
#include "Logger.h"
#include "Statistics.h"
#include "table/SceneDatabase.h"

#include "igd_device.h"
#include "igh_host.h"

#include <cstdlib>

namespace IG {

class HipRenderDevice final : public IRenderDevice {
public:
    explicit HipRenderDevice(const SetupSettings& s)
        : mSetup(s)
    {
        igd_setup setup{};
        setup.gpu_index      = (int32_t)s.target.device();
        setup.acquire_stats  = s.AcquireStats ? 1 : 0;
        setup.debug_trace    = s.DebugTrace ? 1 : 0;
        setup.is_interactive = s.IsInteractive ? 1 : 0;
        // "Normals" / "Albedo" are asked for by name only when the runtime's denoiser is on; keeping them costs two film buffers and
        // one extra camera-ray traversal at iteration 0
        setup.info_aovs      = 1;
        mDev                 = igd_create(&setup);
        if (!mDev) {
            IG_LOG(L_FATAL) << "ig_device_hip: " << igd_last_error() << std::endl;
            std::abort(); // the reference aborts on unrecoverable device errors (Device.cpp:303-306)
        }
    }
    ~HipRenderDevice() override
    {
        igd_destroy(mDev);
        igh_free(mScene);
    }

    // The runtime exports the path of the scene it loaded through IG_HIP_SCENE_FILE (one-line patch in
    // Runtime::loadFromFile); the tables in settings.database stay untouched.
    void assignScene(const SceneSettings&) override
    {
        const char* path = std::getenv("IG_HIP_SCENE_FILE");
        igh_free(mScene);
        mScene = path ? igh_load_file(path, nullptr) : nullptr;
        if (!mScene) {
            IG_LOG(L_ERROR) << "ig_device_hip: " << (path ? igh_last_error() : "IG_HIP_SCENE_FILE is not set") << std::endl;
            return;
        }
        if (igd_assign_scene(mDev, igh_tables(mScene)) != IGD_OK)
            IG_LOG(L_ERROR) << "ig_device_hip: " << igd_last_error() << std::endl;
    }

    void render(const TechniqueVariantShaderSet&, const RenderSettings& rs, ParameterSet* params) override
    {
        if (params) { // the global registry of this iteration (Runtime.cpp:366-387)
            for (const auto& p : params->IntParameters)
                igd_set_parameter_i32(mDev, p.first.c_str(), p.second);
            for (const auto& p : params->FloatParameters)
                igd_set_parameter_f32(mDev, p.first.c_str(), p.second);
            for (const auto& p : params->VectorParameters) {
                const float v[3] = { p.second.x(), p.second.y(), p.second.z() };
                igd_set_parameter_vec3(mDev, p.first.c_str(), v);
            }
        }
        igd_render_settings s{};
        std::vector<float> rays;
        if (rs.rays) { // Runtime::trace: width = #rays, height = 1 (Runtime.cpp:389-446)
            rays.resize(rs.width * 8);
            for (size_t i = 0; i < rs.width; ++i) {
                const Vector3f d = rs.rays[i].Direction.normalized(); // Device.cpp:602-643
                float* o         = rays.data() + i * 8;
                o[0] = rs.rays[i].Origin.x(), o[1] = rs.rays[i].Origin.y(), o[2] = rs.rays[i].Origin.z();
                o[3] = d.x(), o[4] = d.y(), o[5] = d.z();
                o[6] = rs.rays[i].Range.x(), o[7] = rs.rays[i].Range.y();
            }
            s.rays = rays.data();
        }
        s.spi = (int32_t)rs.spi, s.width = (int32_t)rs.width, s.height = (int32_t)rs.height;
        s.iteration = (int32_t)rs.iteration, s.frame = (int32_t)rs.frame, s.user_seed = (int32_t)rs.user_seed;
        s.row_offset = 0, s.row_stride = 1;
        if (igd_render(mDev, &s) != IGD_OK)
            IG_LOG(L_ERROR) << "ig_device_hip: " << igd_last_error() << std::endl;
    }

    void resize(size_t w, size_t h) override { igd_resize(mDev, (int32_t)w, (int32_t)h); }
    void releaseAll() override { igd_release_all(mDev); }

    Target target() const override { return mSetup.target; }
    size_t framebufferWidth() const override { return (size_t)igd_framebuffer_width(mDev); }
    size_t framebufferHeight() const override { return (size_t)igd_framebuffer_height(mDev); }
    bool isInteractive() const override { return mSetup.IsInteractive; }

    AOVAccessor getFramebufferForHost(const std::string& name, bool sync) override
    {
        return AOVAccessor{ const_cast<float*>(igd_framebuffer_host(mDev, name.c_str(), sync ? 1 : 0)) };
    }
    AOVAccessor getFramebufferForDevice(const std::string& name, bool) override { return AOVAccessor{ igd_framebuffer_device(mDev, name.c_str()) }; }
    void clearFramebuffer(const std::string& name) override { igd_clear_framebuffer(mDev, name.c_str()); }
    void clearAllFramebuffer() override { igd_clear_framebuffer(mDev, nullptr); }
    void syncFramebufferHostToDevice(const std::string& name) override
    {
        if (const float* host = igd_framebuffer_host(mDev, name.c_str(), 0))
            igd_sync_framebuffer_to_device(mDev, name.c_str(), host);
    }
    void syncAllFramebufferHostToDevice() override { syncFramebufferHostToDevice({}); }

    // Named buffers, tonemap, imageinfo, bake and runPass are JIT entry points outside the hot path
    // (OUT); the reference's own error behaviour for unknown names is "log + empty".
    size_t getBufferSizeInBytes(const std::string&) override { return 0; }
    bool copyBufferToHost(const std::string&, void*, size_t) override { return false; }
    BufferAccessor getBufferForDevice(const std::string&) override { return BufferAccessor{ nullptr, 0 }; }
    const Statistics* getStatistics() override { return nullptr; }
    void tonemap(uint32_t*, const TonemapSettings&) override { IG_LOG(L_ERROR) << "ig_device_hip: tonemap is not part of the HIP backend" << std::endl; }
    ImageInfoOutput imageinfo(const ImageInfoSettings&) override { return ImageInfoOutput{}; }
    void bake(const ShaderOutput<void*>&, const std::vector<std::string>*, float*) override {}
    void runPass(const ShaderOutput<void*>&) override {}

private:
    SetupSettings mSetup;
    igd_device* mDev  = nullptr;
    igh_scene* mScene = nullptr;
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
