// ParticleSystem.swift -- the reference's tracer class (fluidSolver/Models/ParticleSystem.swift:38-403) with the CPU
// update loop (:242-304) and the vertex stage of its render pass (:341-363 -> particleVertex, Shaders:446-476)
// replaced by calls into libfluidsolver_hip.so.  The tracers live on the device as seven SoA planes.
//
// NOT COMPILED IN THIS REPO'S IMAGE (no Swift toolchain): see Package.swift.
import CFluidSolver

public final class ParticleSystem {
    public var momentum: Float = 0.5            // :63
    public var fluidForce: Float = 0.6          // :70
    public var fadeSpeed: Float = 0.001         // :77 (not read by update(), as in the reference)
    public var spawnCount: Int = 10             // :84
    public var spawnRadius: Float = 15.0        // :91
    public var fluidDecayRate: Float = 3.0      // :98
    public var linkToViscosity: Bool = true     // :104
    public var particleSize: Float = 1.0        // :111
    public var isEnabled: Bool = true           // :118

    private let solver: FluidSolver
    private var rng: SystemRandomNumberGenerator = SystemRandomNumberGenerator()

    public init(fluidSolver: FluidSolver) { solver = fluidSolver }

    public var particleCount: Int { Int(fs_particles_count(solver.context)) }   // :116

    private func check(_ rc: Int32) throws {
        if rc != FS_OK { throw FluidSolverError(code: rc, description: String(cString: fs_last_error(solver.context))) }
    }

    /// Replace the tracer set (seven arrays of equal length: px py vx vy life mass age).
    public func setParticles(px: [Float], py: [Float], vx: [Float], vy: [Float], life: [Float], mass: [Float],
                             age: [Float]) throws {
        try check(fs_particles_set(solver.context, px.count, px, py, vx, vy, life, mass, age))
    }
    public func getParticles() throws -> [[Float]] {
        let n = particleCount
        var a = [[Float]](repeating: [Float](repeating: 0, count: n), count: 7)
        try check(fs_particles_get(solver.context, n, &a[0], &a[1], &a[2], &a[3], &a[4], &a[5], &a[6]))
        return a
    }

    /// addParticles(at:count:windowSize:), :186-231, on the device (fs_particles_spawn): live tracers kept in age order,
    /// oldest dropped at the cap (:196-199).  The seed stands in for Float.random's system RNG (:203-212).
    public func addParticles(atX x: Float, y: Float, count: Int? = nil) throws {
        guard isEnabled else { return }
        let seed = UInt32.random(in: 1...UInt32.max, using: &rng)
        try check(fs_particles_spawn(solver.context, x, y, count ?? spawnCount, spawnRadius, seed))
    }

    /// update(fluidSolver:windowSize:), :242-304, on the device, sampling the real velocity field.
    public func update(windowWidth: Float, windowHeight: Float) throws {
        guard isEnabled, particleCount > 0 else { return }                                 // :243
        var pp = fs_particle_params(momentum: momentum, fluidForce: fluidForce, fluidDecayRate: fluidDecayRate,
                                    linkToViscosity: linkToViscosity ? 1 : 0, viscosity: solver.viscosity,
                                    windowW: windowWidth, windowH: windowHeight)
        try check(fs_particles_update(solver.context, &pp))
    }

    /// What render(in:viewportSize:) draws (:341-363): per tracer the line (pos - vel * particleSize, pos) in NDC as
    /// (x0, y0, x1, y1), and alpha = life -- ready for a line-list vertex buffer.
    public func lineVertices(viewportWidth: Float, viewportHeight: Float) throws -> (xy: [Float], alpha: [Float]) {
        let n = particleCount
        var xy = [Float](repeating: 0, count: 4 * n), alpha = [Float](repeating: 0, count: n)
        try check(fs_particle_lines(solver.context, viewportWidth, viewportHeight, particleSize, &xy, &alpha))
        return (xy, alpha)
    }

    public func reset() throws {                                                           // :366-370
        try setParticles(px: [], py: [], vx: [], vy: [], life: [], mass: [], age: [])
    }
}
