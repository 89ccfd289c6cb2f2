// debug.h -- ONE switchboard for tests and tuning experiments: the environment variable
//   POINTOPS_DEBUG="key=value,key=value,..."
// read at call time (one getenv per lookup, only on host launch paths).  Unset = every knob at its
// default; no knob ever changes results, only which exact path computes them.
//   knn_generic=1      knn_wide's plain generic kernel instead of its LDS-tiled forms
//   knn_small=0|1      few queries: never / always the wave-per-query kernel (knn_small.hip; default: by shape)
//   knn_small_q=1|2    that kernel: one query per wave / as many as the list size allows (default: by query count)
//   grid_quad=0|1      force the quad pass of the grid KNN off / on (default: by the batch's query count)
//   grid_long_box=0|1  K in (32, 64]: uncertified queries to the wave search / to the box search (default: by query count)
//   grid_refine=0      no refined cells (over-full neighbourhoods still go to the box search, over whole cells)
//   grid_same=0        do not reuse the point sort as the query order when p1 is p2
//   grid_c_scale=F     multiply the grid KNN's points-per-cell target (sweeps)
//   ball_small=0|1     ball query, few queries: never / always the wave-per-query kernel (ball_small.hip; default: by shape)
//   ball_grid=0|1      ball query: never / whenever possible through the grid (default: by shape)
//   ball_factor=F      ball query: grid-or-scan crossover constant
//   ball_order=0       ball query: scan-mode clouds keep their queries in storage order (no coarse-cell order)
//   knn_bwd_mode=a|t   knn backward grad_p2: device atomics / LDS tiles;  knn_bwd_split=S
//   gather_bwd_mode=a|t, gather_bwd_split=S   the same for knn_gather's backward
//   chamfer_overlap=0  one-call chamfer: the reverse search on the caller's stream instead of a side stream
//   fps_small=0        FPS: clouds of up to 4096 points through the 16-wave cluster kernel instead of the 4-wave one
//   fps_small_ppt=0    FPS clusters: never four points per lane for clouds above 4096 points;  fps_ppt=4|8|16 forces it
//   fps_mode=0|1|2     FPS clusters: round-robin members / XCD-local members / XCD-local + L2 exchange
//   fps_spin_limit=N   FPS exchange spin bound (tests force the timeout repair path with 0)
#pragma once

namespace pointops {

// integer knob (value parsed with strtol), `dflt` when absent
long debug_knob(const char* key, long dflt);
// floating-point knob
double debug_knob_f(const char* key, double dflt);
// first character of the knob's value, 0 when absent
char debug_knob_c(const char* key);

}  // namespace pointops
