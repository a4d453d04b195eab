// (28 is the value of the reference's own demo, src/main.rs:13-48; 31 the benchmark's; 10 that of its known-answer test, tests/main.rs:41-57;
// 5, 7, 11, 17, 25, 31 are
// the six of its stress grid, tests/main.rs:82-89.)
// Minimizer lengths l that get a fully unrolled, compile-time instantiation of the tiled kernel (one translation unit
// each, built from s2k_tile_inst.hip with -DS2K_TILE_L=<l>; keep STATIC_LS in the Makefile in step).  Every other
// l <= 64 runs the same kernel with a run-time l (about 1.4x slower hash loop).
#pragma once
#define S2K_STATIC_LS(X) X(5) X(7) X(10) X(11) X(12) X(15) X(17) X(21) X(25) X(28) X(31)
