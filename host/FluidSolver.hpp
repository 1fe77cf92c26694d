// FluidSolver.hpp -- C++ host-side mirror of the reference's Swift classes over the C ABI.
//
// The reference host is Swift (fluidSolver/Models/FluidSolver.swift, ParticleSystem.swift); no Swift
// toolchain exists in this image, so the host side above include/fluidsolver.h is written in C++ with
// the same class names, method names, parameter names and defaults.  All compute happens in
// libfluidsolver_hip.so; these classes only hold the host state the Swift classes hold (parameters, the
// previous touch position).  Header-only: #include and link -lfluidsolver_hip.
#pragma once
#include <cmath>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

#include "fluidsolver.h"

namespace fluidsolver {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };

// FluidPreset, FluidPresets.swift:10-23
struct FluidPreset {
    const char *name;
    float viscosity, diffusion, deltaTime, vorticityStrength, fadeSpeed;
    int solverIterations;
    float forceMultiplier, colorMultiplier, brightness;
};

// FluidPresets.swift:27-177
inline const std::vector<FluidPreset> &presets()
{
    static const std::vector<FluidPreset> p = {
        {"Default", 0.00001f, 0.0f, 0.59f, 0.0f, 0.005f, 15, 24.0f, 100.0f, 0.06f},
        {"Viscous", 0.001f, 0.0f, 0.5f, 0.0f, 0.003f, 20, 30.0f, 80.0f, 0.08f},
        {"Turbulent", 0.00001f, 0.0f, 0.8f, 25.0f, 0.002f, 10, 50.0f, 120.0f, 0.1f},
        {"Smoke", 0.00005f, 0.00001f, 0.4f, 10.0f, 0.008f, 15, 35.0f, 60.0f, 0.15f},
        {"Water", 0.000001f, 0.0f, 1.0f, 5.0f, 0.001f, 25, 40.0f, 90.0f, 0.07f},
        {"Ink", 0.0001f, 0.00005f, 0.6f, 15.0f, 0.004f, 18, 45.0f, 150.0f, 0.12f},
        {"Fast Flow", 0.000001f, 0.0f, 2.0f, 8.0f, 0.01f, 8, 60.0f, 80.0f, 0.05f},
        {"High Detail", 0.00001f, 0.0f, 0.5f, 12.0f, 0.003f, 30, 30.0f, 100.0f, 0.08f},
        {"Paint", 0.0005f, 0.0001f, 0.3f, 0.0f, 0.0005f, 20, 20.0f, 200.0f, 0.2f},
        {"Minimal", 0.0f, 0.0f, 0.5f, 0.0f, 0.02f, 5, 50.0f, 50.0f, 0.1f},
    };
    return p;
}

// ColorPalette.rainbow, ColorPalette.swift:44 + hsbToRgb :107-128
inline float4 hsbToRgb(float h, float s, float b)
{
    float c = b * s;
    float x = c * (1.0f - std::fabs(std::fmod(h * 6.0f, 2.0f) - 1.0f));
    float m = b - c;
    float r, g, bl;
    if (h < 1.0f / 6.0f) { r = c; g = x; bl = 0; }
    else if (h < 2.0f / 6.0f) { r = x; g = c; bl = 0; }
    else if (h < 3.0f / 6.0f) { r = 0; g = c; bl = x; }
    else if (h < 4.0f / 6.0f) { r = 0; g = x; bl = c; }
    else if (h < 5.0f / 6.0f) { r = x; g = 0; bl = c; }
    else { r = c; g = 0; bl = x; }
    return {r + m, g + m, bl + m, 1.0f};
}

// class FluidSolver, FluidSolver.swift:17-673
class FluidSolver {
  public:
    enum class DisplayMode { density, velocity, speed, vorticity, particlesOnly, none };  // :214-225 (+ headless)

    // @Published parameters with the defaults of :139-233
    float viscosity = 0.00015f, diffusion = 0.0f, deltaTime = 0.5f, vorticityStrength = 0.0f, fadeSpeed = 0.002f;
    int solverIterations = 10;
    float forceMultiplier = 24.0f, colorMultiplier = 100.0f, brightness = 0.06f, forceRadius = 5.0f;
    DisplayMode displayMode = DisplayMode::density;

    explicit FluidSolver(int gridWidth = 512, int gridHeight = 512, int device = 0, long maxParticles = 0,
                         int resolutionScale = 1)
        : gridWidth_(gridWidth), gridHeight_(gridHeight), scale_(resolutionScale), device_(device),
          maxParticles_(maxParticles)
    {
        setupTextures();
    }
    ~FluidSolver() { fs_destroy(ctx_); }
    FluidSolver(const FluidSolver &) = delete;
    FluidSolver &operator=(const FluidSolver &) = delete;

    int scaledWidth() const { return std::max(32, gridWidth_ / scale_); }    // :99-101
    int scaledHeight() const { return std::max(32, gridHeight_ / scale_); }  // :104-106

    void setResolution(int resolution)  // :252-255 -> reinitialize() :260-263
    {
        if (resolution == gridWidth_ && resolution == gridHeight_) return;
        gridWidth_ = gridHeight_ = resolution;
        setupTextures();
        reset();
    }
    void reset()  // :371-374
    {
        check(fs_reset(ctx_));
        hasPrevious_ = false;
    }
    void applyPreset(const FluidPreset &p)  // :382-399
    {
        viscosity = p.viscosity; diffusion = p.diffusion; deltaTime = p.deltaTime;
        vorticityStrength = p.vorticityStrength; fadeSpeed = p.fadeSpeed; solverIterations = p.solverIterations;
        forceMultiplier = p.forceMultiplier; colorMultiplier = p.colorMultiplier; brightness = p.brightness;
    }
    // addForce(at:windowSize:), :409-428; `time` replaces the wall-clock palette phase of :422-423
    void addForce(float2 at, float2 windowSize, float time)
    {
        // CGPoint/CGSize are double in the reference: Float(position.x / windowSize.width) * Float(scaledWidth)
        float2 pos = {(float)((double)at.x / (double)windowSize.x) * (float)scaledWidth(),
                      (float)((double)at.y / (double)windowSize.y) * (float)scaledHeight()};
        float2 force = {0, 0};
        if (hasPrevious_) force = {(pos.x - previous_.x) * forceMultiplier, (pos.y - previous_.y) * forceMultiplier};
        float4 c = hsbToRgb(time, 1.0f, 1.0f);
        applyForce(pos, force, {c.x * colorMultiplier, c.y * colorMultiplier, c.z * colorMultiplier, c.w * colorMultiplier});
        previous_ = pos;
        hasPrevious_ = true;
    }
    void endInteraction() { hasPrevious_ = false; }  // :431-433
    void applyForce(float2 position, float2 force, float4 color)  // :444-470 (grid units)
    {
        const float col[4] = {color.x, color.y, color.z, color.w};
        check(fs_add_force(ctx_, position.x, position.y, force.x, force.y, col, forceRadius, deltaTime));
    }
    void update()  // :477-627
    {
        fs_params p;
        p.viscosity = viscosity; p.diffusion = diffusion; p.deltaTime = deltaTime;
        p.vorticityStrength = vorticityStrength; p.fadeSpeed = fadeSpeed; p.solverIterations = solverIterations;
        p.displayMode = displayMode == DisplayMode::none ? FS_DISPLAY_NONE : (int)displayMode;
        p.brightness = brightness;
        check(fs_set_params(ctx_, &p));
        check(fs_step(ctx_));
    }
    void sync() { check(fs_sync(ctx_)); }
    // texture (:630-632): the display field, planar RGBA [4][H][W]
    std::vector<float> texture() { return readField(FS_FIELD_DISPLAY, 4); }
    std::vector<float> readField(int field, int planes)
    {
        std::vector<float> out((size_t)planes * scaledWidth() * scaledHeight());
        check(fs_read_field(ctx_, field, out.data(), out.size() * sizeof(float)));
        return out;
    }
    // getVelocityAt(normalizedPos:), :648-663, against the real field (debug helper; tracers do this on device)
    float2 getVelocityAt(float2 n)
    {
        const int W = scaledWidth(), H = scaledHeight();
        int x = (int)std::fmax(0.0f, std::fmin((float)(W - 1), n.x * (float)W));
        int y = (int)std::fmax(0.0f, std::fmin((float)(H - 1), n.y * (float)H));
        std::vector<float> v = readField(FS_FIELD_VELOCITY, 2);
        return {v[(size_t)y * W + x] / (float)W, v[(size_t)W * H + (size_t)y * W + x] / (float)H};
    }
    fs_ctx *ctx() { return ctx_; }

  private:
    void setupTextures()  // :322-343
    {
        fs_destroy(ctx_);
        ctx_ = nullptr;
        check(fs_create(scaledWidth(), scaledHeight(), device_, maxParticles_, &ctx_));
    }
    void check(int rc)
    {
        if (rc != FS_OK) throw Error(rc, fs_last_error(ctx_));
    }
    int gridWidth_, gridHeight_, scale_, device_;
    long maxParticles_;
    fs_ctx *ctx_ = nullptr;
    bool hasPrevious_ = false;
    float2 previous_{0, 0};
};

// class ParticleSystem, ParticleSystem.swift:38-403 (update on the device)
class ParticleSystem {
  public:
    float momentum = 0.5f, fluidForce = 0.6f, fluidDecayRate = 3.0f;  // Part:63,70,98
    bool linkToViscosity = true;                                       // Part:104
    explicit ParticleSystem(FluidSolver &solver) : s_(solver) {}
    void setParticles(const std::vector<float> &px, const std::vector<float> &py, const std::vector<float> &vx,
                      const std::vector<float> &vy, const std::vector<float> &life, const std::vector<float> &mass,
                      const std::vector<float> &age)
    {
        int rc = fs_particles_set(s_.ctx(), (long)px.size(), px.data(), py.data(), vx.data(), vy.data(), life.data(),
                                  mass.data(), age.data());
        if (rc) throw Error(rc, fs_last_error(s_.ctx()));
    }
    float spawnRadius = 15.0f;  // Part:91
    int spawnCount = 10;        // Part:84
    // addParticles(at:count:windowSize:), Part:186-231, on the device; `seed` stands in for the system RNG (:203-212)
    void addParticles(float2 at, int count, unsigned seed)
    {
        int rc = fs_particles_spawn(s_.ctx(), at.x, at.y, count, spawnRadius, seed);
        if (rc) throw Error(rc, fs_last_error(s_.ctx()));
    }
    void update(float2 windowSize)  // update(fluidSolver:windowSize:), Part:242-304
    {
        fs_particle_params pp{momentum, fluidForce, fluidDecayRate, linkToViscosity ? 1 : 0, s_.viscosity,
                              windowSize.x, windowSize.y};
        int rc = fs_particles_update(s_.ctx(), &pp);
        if (rc) throw Error(rc, fs_last_error(s_.ctx()));
    }
    long particleCount() { return fs_particles_count(s_.ctx()); }
    float particleSize = 1.0f;  // Part:111
    // What render(in:viewportSize:) draws (Part:341-363 -> particleVertex, Shaders:446-476): per tracer the line
    // (pos - vel * particleSize, pos) in NDC as (x0, y0, x1, y1), and alpha = life.
    void lineVertices(float2 viewportSize, std::vector<float> &xy4, std::vector<float> &alpha)
    {
        const size_t n = (size_t)particleCount();
        xy4.resize(4 * n);
        alpha.resize(n);
        int rc = fs_particle_lines(s_.ctx(), viewportSize.x, viewportSize.y, particleSize, xy4.data(), alpha.data());
        if (rc) throw Error(rc, fs_last_error(s_.ctx()));
    }

  private:
    FluidSolver &s_;
};

}  // namespace fluidsolver
