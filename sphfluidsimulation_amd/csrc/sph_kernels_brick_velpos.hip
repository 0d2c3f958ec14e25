// VelPos.compute::CSMain replacement: LDS-tiled brick kernel, see sph_brick_impl.inc.
// Brick shape and tile size tuned on MI355X at 16.7 M particles (profiles/): 4x4x4 cells, halo 2,
// 256 threads, 2304-particle tile + 32-entry survivor lists -> 3 bricks resident per CU.
#define SPH_BRICK_NS brick_velpos
#define SPH_BRICK_EMIT_VELPOS 1
#ifndef SPH_V_BX
#define SPH_V_BX 4
#endif
#ifndef SPH_V_T
#define SPH_V_T 256
#endif
#ifndef SPH_V_CAP
#define SPH_V_CAP 2304
#endif
#ifndef SPH_VELPOS_MINWAVES
#define SPH_VELPOS_MINWAVES 3
#endif
#define SPH_BRICK_BX SPH_V_BX
#define SPH_BRICK_T SPH_V_T
#define SPH_BRICK_CAP SPH_V_CAP
#include "sph_brick_impl.inc"
