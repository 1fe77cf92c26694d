// swift-tools-version:5.7
// Swift host package over the C ABI of include/fluidsolver.h (SURVEY.md 8(f) rank 2).
// NOT COMPILED IN THIS REPO'S IMAGE: the image has no Swift toolchain (swift / swiftc not found), so this package
// is written against the header and has not been built or run here; the C++ mirror (host/FluidSolver.hpp) and the
// Python mirror (fluidsolver_amd/solver.py) are the hosts the tests exercise.  Build where Swift exists with
//   swift build -Xlinker -L<repo>/fluidsolver_amd -Xcc -I<repo>/include
import PackageDescription

let package = Package(
    name: "FluidSolverHIP",
    products: [.library(name: "FluidSolverCore", targets: ["FluidSolverCore"])],
    targets: [
        .systemLibrary(name: "CFluidSolver", path: "Sources/CFluidSolver"),
        .target(name: "FluidSolverCore", dependencies: ["CFluidSolver"]),
    ]
)
