// Density.compute::CSMain replacement: LDS-tiled brick kernel, see sph_brick_impl.inc.
// Brick shape and tile size tuned on MI355X at 16.7 M particles (scripts/gpu_variants2.sh): 4x4x4 cells, halo 1,
// 256 threads, 1024-particle tile -> 7 bricks resident per CU (6 for the instantiation that records neighbour lists:
// + the slot table).  (8x4x4 cells with a 2048 / 1792-particle tile: 1.71 / 2.37 ms instead of 1.57 / 2.22.)
#define SPH_BRICK_NS brick_density
#ifndef SPH_DENSITY_MINWAVES
#define SPH_DENSITY_MINWAVES 7
#endif
#ifndef SPH_RECORD_CAP
#define SPH_RECORD_CAP 928    /* tile of the recording pass and of k_velpos_tile (32 B per particle there: five workgroups per CU) */
#endif
#ifndef SPH_RECORD_MINWAVES
#define SPH_RECORD_MINWAVES 5     /* exact mode: 32 staged candidates per thread = 16 KB, five workgroups per CU */
#endif
#define SPH_BRICK_EMIT_DENSITY 1
#ifndef SPH_SCAN_UNROLL
#define SPH_SCAN_UNROLL 2     /* slots per trip of the density scan (2 / 3 / 4: 2.36 / 2.36 / 2.41 ms with RECORD: the runs are short, ~11 slots) */
#endif
#ifndef SPH_D_BX
#define SPH_D_BX 4
#endif
#ifndef SPH_D_T
#define SPH_D_T 256
#endif
#ifndef SPH_D_CAP
#define SPH_D_CAP 1024
#endif
#define SPH_BRICK_BX SPH_D_BX
#define SPH_BRICK_T SPH_D_T
#define SPH_BRICK_CAP SPH_D_CAP
#include "sph_brick_impl.inc"
