/*
 * fdwave_compat.h -- the reference's own propagation entry points (cuda_reference_RTM), same names,
 * same argument lists, implemented on libfdwave.  Link libfdwave_rtm_compat.so INSTEAD of compiling the
 * kernels + fd_* functions of src/fd-code.cu (lines 53-341) and main() works unchanged.
 *
 * Prototypes: fd_init is published in cuda_reference_RTM/lib/include/functions.h:15; fd_forward and
 * fd_back are defined in src/fd-code.cu:247 and :290 (note the nz-before-nx order of the raw sizes).
 * Arrays are the reference's SU-style float** whose [0] is one contiguous block (functions.c:203-217).
 * State is process-global like the reference's file-scope device pointers (fd-code.cu:31-33); errors
 * (the reference reports none) are printed to stderr and abort the process.
 */
#ifndef FDWAVE_COMPAT_H
#define FDWAVE_COMPAT_H
#ifdef __cplusplus
extern "C" {
#endif

void fd_init(int order, int nxe, int nze, int nxb, int nzb, int nt, int ns, float fac, float dx, float dz, float dt);
void fd_forward(int order, float **p, float **pp, float **v2, int nze, int nxe, int nt, int is, int sz, int *sx,
                float *srce, int propag);
void fd_back(int order, float **p, float **pp, float **pr, float **ppr, float **v2, int nze, int nxe, int nt, int is,
             int sz, int gz, float ***snaps, float **imloc, float **d_obs);
void fd_free(void); /* replaces the cudaFree block of main(), fd-code.cu:569-582 */

#ifdef __cplusplus
}
#endif
#endif
