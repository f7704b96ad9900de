/* lama_host.h -- C surface of the host-side library (liblama_host.so).  WORK IN PROGRESS header;
 * see the full documentation block below once the PF facade lands. */
#ifndef LAMA_HOST_H
#define LAMA_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Seeded synthetic corridor log.  pts: (steps+1) x beams x 3 doubles (sensor frame),
 * odom_xyr / truth_xyr: (steps+1) x 3 doubles (x, y, yaw).  truth_xyr may be NULL.  Returns 0. */
int lama_corridor_generate(int steps, int beams, double* pts, double* odom_xyr, double* truth_xyr);

#ifdef __cplusplus
}
#endif
#endif
