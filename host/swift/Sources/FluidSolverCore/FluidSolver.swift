// FluidSolver.swift -- the reference's host class (fluidSolver/Models/FluidSolver.swift:22-673) with its Metal
// half replaced by calls into libfluidsolver_hip.so.  Same class name, property names, defaults and method names, so
// the views and gesture handlers of the app keep compiling; SwiftUI's @Published wrappers are left to the app layer
// (plain stored properties here: the package has no UI dependency).  Line numbers cite the reference.
//
// NOT COMPILED IN THIS REPO'S IMAGE (no Swift toolchain): see Package.swift.
import CFluidSolver

public struct FluidSolverError: Error, CustomStringConvertible {
    public let code: Int32
    public let description: String
}

public enum DisplayMode: Int32 {           // :214-225; .none = headless, no render pass
    case none = -1, density = 0, velocity = 1, speed = 2, vorticity = 3, particlesOnly = 4
}

public final class FluidSolver {
    // ---- parameters update() reads (:139-202, :227) ----
    public var viscosity: Float = 0.00015
    public var diffusion: Float = 0.0
    public var deltaTime: Float = 0.5
    public var vorticityStrength: Float = 0.0
    public var fadeSpeed: Float = 0.002
    public var solverIterations: Int = 10
    public var forceMultiplier: Float = 24.0    // :181
    public var colorMultiplier: Float = 100.0   // :188
    public var brightness: Float = 0.06         // :195
    public var forceRadius: Float = 5.0         // :202
    public var displayMode: DisplayMode = .density

    public private(set) var gridWidth: Int
    public private(set) var gridHeight: Int
    public var resolutionScale: Int = 1 { didSet { if resolutionScale != oldValue { reinitialize() } } }   // :89-96
    public var scaledWidth: Int { max(32, gridWidth / max(1, resolutionScale)) }      // :99-102
    public var scaledHeight: Int { max(32, gridHeight / max(1, resolutionScale)) }    // :103-106

    private var ctx: OpaquePointer?             // replaces device / commandQueue / library / 13 pipelines / 9 textures
    private let device: Int32
    private let maxParticles: Int
    private var previousTouchPosition: SIMD2<Float>?

    public init(gridWidth: Int = 512, gridHeight: Int? = nil, device: Int32 = 0, maxParticles: Int = 0) throws {   // :240-244
        self.gridWidth = gridWidth
        self.gridHeight = gridHeight ?? gridWidth
        self.device = device
        self.maxParticles = maxParticles
        try setupTextures()
    }
    deinit { fs_destroy(ctx) }

    /// The opaque context, for ParticleSystem and for hosts that call the per-kernel layer themselves.
    public var context: OpaquePointer? { ctx }

    private func check(_ rc: Int32) throws {
        if rc != FS_OK { throw FluidSolverError(code: rc, description: String(cString: fs_last_error(ctx))) }
    }
    private func setupTextures() throws {       // :322-343: allocate the fields of the scaled grid, all zero
        fs_destroy(ctx)
        ctx = nil
        try check(fs_create(Int32(scaledWidth), Int32(scaledHeight), device, maxParticles, &ctx))
    }
    private func reinitialize() {               // :260-263
        try? setupTextures()
        previousTouchPosition = nil
    }

    public func setResolution(_ resolution: Int) {   // :252-255
        gridWidth = resolution
        gridHeight = resolution
        reinitialize()
    }
    public func reset() {                       // :371-374 -> clearTextures() :348-366
        _ = fs_reset(ctx)
        previousTouchPosition = nil
    }
    public func applyPreset(_ p: FluidPreset) { // :382-399 (the grid resolution is deliberately left alone)
        viscosity = p.viscosity; diffusion = p.diffusion; deltaTime = p.deltaTime
        vorticityStrength = p.vorticityStrength; fadeSpeed = p.fadeSpeed; solverIterations = p.solverIterations
        forceMultiplier = p.forceMultiplier; colorMultiplier = p.colorMultiplier; brightness = p.brightness
    }

    /// addForce(at:windowSize:), :409-428.  `paletteTime` in [0,1) replaces the wall-clock phase of :422-423 so that
    /// runs are reproducible; pass the app's own palette colour with applyForce if it has one.
    public func addForce(atX x: Double, y: Double, windowWidth: Double, windowHeight: Double, paletteTime: Float) throws {
        let pos = SIMD2<Float>(Float(x / windowWidth) * Float(scaledWidth), Float(y / windowHeight) * Float(scaledHeight))
        var force = SIMD2<Float>(0, 0)
        if let prev = previousTouchPosition { force = (pos - prev) * forceMultiplier }   // :417
        let c = ColorPalette.hsbToRgb(h: paletteTime, s: 1, b: 1) * colorMultiplier       // :422-424 (rainbow)
        try applyForce(position: pos, force: force, color: c)
        previousTouchPosition = pos
    }
    public func endInteraction() { previousTouchPosition = nil }   // :431-433

    /// applyForce(position:force:color:), :444-470, in grid units: the five arguments of the addForces kernel.
    public func applyForce(position: SIMD2<Float>, force: SIMD2<Float>, color: SIMD4<Float>) throws {
        var c = [color.x, color.y, color.z, color.w]
        try check(fs_add_force(ctx, position.x, position.y, force.x, force.y, &c, forceRadius, deltaTime))
    }

    /// update(), :477-627: one call instead of 2N+7 dispatches; asynchronous like commandBuffer.commit() (:626).
    public func update() throws {
        var p = fs_params(viscosity: viscosity, diffusion: diffusion, deltaTime: deltaTime,
                          vorticityStrength: vorticityStrength, fadeSpeed: fadeSpeed,
                          solverIterations: Int32(solverIterations), displayMode: displayMode.rawValue,
                          brightness: brightness)
        try check(fs_set_params(ctx, &p))
        try check(fs_step(ctx))
    }
    public func sync() throws { try check(fs_sync(ctx)) }

    /// The display texture (:630-632) as planar RGBA fp32 [4][H][W], for whatever presents frames.
    public func texture() throws -> [Float] { try readField(FS_FIELD_DISPLAY, planes: 4) }
    public func readField(_ field: Int32, planes: Int) throws -> [Float] {
        var out = [Float](repeating: 0, count: planes * scaledWidth * scaledHeight)
        try check(fs_read_field(ctx, field, &out, out.count * MemoryLayout<Float>.size))
        return out
    }
    /// getVelocityAt(normalizedPos:), :648-663, against the real field (the reference's cache is never filled,
    /// :639, :669-672).  A debugging aid: tracers sample on the device (ParticleSystem.update).
    public func getVelocityAt(normalizedPos n: SIMD2<Float>) throws -> SIMD2<Float> {
        let w = scaledWidth, h = scaledHeight
        let x = Int(max(0, min(Float(w - 1), n.x * Float(w)))), y = Int(max(0, min(Float(h - 1), n.y * Float(h))))
        let v = try readField(FS_FIELD_VELOCITY, planes: 2)
        return SIMD2<Float>(v[y * w + x] / Float(w), v[w * h + y * w + x] / Float(h))
    }
}
