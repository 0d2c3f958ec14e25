// Density.compute::CSMain replacement: LDS-tiled brick kernel, see sph_brick_impl.inc.
// Brick shape and tile size tuned on MI355X at 16.7 M particles (profiles/): 8x4x4 cells, halo 1,
// 256 threads, 2048-particle tile -> 5 bricks resident per CU.
#define SPH_BRICK_NS brick_density
#define SPH_BRICK_EMIT_DENSITY 1
#ifndef SPH_D_BX
#define SPH_D_BX 8
#endif
#ifndef SPH_D_T
#define SPH_D_T 256
#endif
#ifndef SPH_D_CAP
#define SPH_D_CAP 2048
#endif
#define SPH_BRICK_BX SPH_D_BX
#define SPH_BRICK_T SPH_D_T
#define SPH_BRICK_CAP SPH_D_CAP
#include "sph_brick_impl.inc"
