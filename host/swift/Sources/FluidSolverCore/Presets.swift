// Presets.swift -- the ten parameter presets (fluidSolver/Models/FluidPresets.swift:27-177) and the rainbow palette's
// HSB conversion (fluidSolver/Models/ColorPalette.swift:107-128) as data for the host package.  The numbers are the
// reference's; fluidsolver_amd/presets.py holds the same table and all eight palettes, with tests.
//
// NOT COMPILED IN THIS REPO'S IMAGE (no Swift toolchain): see Package.swift.

public struct FluidPreset {
    public let name: String
    public let viscosity, diffusion, deltaTime, vorticityStrength, fadeSpeed: Float
    public let solverIterations: Int
    public let forceMultiplier, colorMultiplier, brightness: Float

    public static let all: [FluidPreset] = [
        FluidPreset(name: "Default", viscosity: 0.00001, diffusion: 0, deltaTime: 0.59, vorticityStrength: 0, fadeSpeed: 0.005, solverIterations: 15, forceMultiplier: 24, colorMultiplier: 100, brightness: 0.06),
        FluidPreset(name: "Viscous", viscosity: 0.001, diffusion: 0, deltaTime: 0.5, vorticityStrength: 0, fadeSpeed: 0.003, solverIterations: 20, forceMultiplier: 30, colorMultiplier: 80, brightness: 0.08),
        FluidPreset(name: "Turbulent", viscosity: 0.00001, diffusion: 0, deltaTime: 0.8, vorticityStrength: 25, fadeSpeed: 0.002, solverIterations: 10, forceMultiplier: 50, colorMultiplier: 120, brightness: 0.1),
        FluidPreset(name: "Smoke", viscosity: 0.00005, diffusion: 0.00001, deltaTime: 0.4, vorticityStrength: 10, fadeSpeed: 0.008, solverIterations: 15, forceMultiplier: 35, colorMultiplier: 60, brightness: 0.15),
        FluidPreset(name: "Water", viscosity: 0.000001, diffusion: 0, deltaTime: 1.0, vorticityStrength: 5, fadeSpeed: 0.001, solverIterations: 25, forceMultiplier: 40, colorMultiplier: 90, brightness: 0.07),
        FluidPreset(name: "Ink", viscosity: 0.0001, diffusion: 0.00005, deltaTime: 0.6, vorticityStrength: 15, fadeSpeed: 0.004, solverIterations: 18, forceMultiplier: 45, colorMultiplier: 150, brightness: 0.12),
        FluidPreset(name: "Fast Flow", viscosity: 0.000001, diffusion: 0, deltaTime: 2.0, vorticityStrength: 8, fadeSpeed: 0.01, solverIterations: 8, forceMultiplier: 60, colorMultiplier: 80, brightness: 0.05),
        FluidPreset(name: "High Detail", viscosity: 0.00001, diffusion: 0, deltaTime: 0.5, vorticityStrength: 12, fadeSpeed: 0.003, solverIterations: 30, forceMultiplier: 30, colorMultiplier: 100, brightness: 0.08),
        FluidPreset(name: "Paint", viscosity: 0.0005, diffusion: 0.0001, deltaTime: 0.3, vorticityStrength: 0, fadeSpeed: 0.0005, solverIterations: 20, forceMultiplier: 20, colorMultiplier: 200, brightness: 0.2),
        FluidPreset(name: "Minimal", viscosity: 0, diffusion: 0, deltaTime: 0.5, vorticityStrength: 0, fadeSpeed: 0.02, solverIterations: 5, forceMultiplier: 50, colorMultiplier: 50, brightness: 0.1),
    ]
}

public enum ColorPalette {
    /// hsbToRgb, ColorPalette.swift:107-128
    public static func hsbToRgb(h: Float, s: Float, b: Float) -> SIMD4<Float> {
        let c = b * s
        let x = c * (1 - abs((h * 6).truncatingRemainder(dividingBy: 2) - 1))
        let m = b - c
        let rgb: SIMD3<Float>
        switch h {
        case ..<(1.0 / 6.0): rgb = SIMD3(c, x, 0)
        case ..<(2.0 / 6.0): rgb = SIMD3(x, c, 0)
        case ..<(3.0 / 6.0): rgb = SIMD3(0, c, x)
        case ..<(4.0 / 6.0): rgb = SIMD3(0, x, c)
        case ..<(5.0 / 6.0): rgb = SIMD3(x, 0, c)
        default: rgb = SIMD3(c, 0, x)
        }
        return SIMD4<Float>(rgb.x + m, rgb.y + m, rgb.z + m, 1)
    }
}
