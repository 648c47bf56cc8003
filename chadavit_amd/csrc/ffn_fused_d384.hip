// ChAda-ViT-Small (D = 384) build of the fused FFN kernels: ffn_fused.hip compiled with FFN_FD = 384, eight waves of one 16-row tile
// each (see the note at the top of that file).  Exports chada_int_*_d384, reached through the public entry points' dispatch on D.
#undef FFN_NW
#undef FFN_RT
#undef FFN_MIN_WAVES
#define FFN_FD 384
#define FFN_NW 8
#define FFN_RT 1
#include "ffn_fused.hip"
