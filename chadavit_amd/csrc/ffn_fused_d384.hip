// ChAda-ViT-Small (D = 384) build of the fused FFN kernels: ffn_fused.hip compiled with FFN_FD = 384, eight waves of one 16-row tile
// each (see the note at the top of that file).  Exports chada_int_*_d384, reached through the public entry points' dispatch on D.
// (FFN384_NW / FFN384_RT / FFN384_MIN_WAVES: the tiling of THIS build for side builds; the command line's FFN_NW / FFN_RT belong to the D = 192 build.)
#undef FFN_NW
#undef FFN_RT
#undef FFN_MIN_WAVES
#define FFN_FD 384
#ifdef FFN384_NW
#define FFN_NW FFN384_NW
#define FFN_RT FFN384_RT
#ifdef FFN384_MIN_WAVES
#define FFN_MIN_WAVES FFN384_MIN_WAVES
#endif
#else
#define FFN_NW 8
#define FFN_RT 1
#endif
#include "ffn_fused.hip"
