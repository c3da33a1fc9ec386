/* slice3d_mesh.h — C ABI of libslice3d_mesh.so (slice3d_amd/csrc_mesh/mesh.cpp): host C++ twins of the two native
 * utilities reg_slices/reconstruct.py reaches.  The device versions (the default path of
 * slice3d_amd.generator.Generator3D) are declared in slice3d_hip.h; these keep the reference's exact point ORDER and
 * serve hosts without a GPU buffer at hand (tests, tools).  Python binding: slice3d_amd/mesh.py.
 *
 *   MISE            reference reg_slices/src_convonet/utils/libmise/mise.pyx:33-368
 *   marching cubes  reference libmcubes/pywrapper.cpp:90-127 -> marchingcubes.h:23-193
 */
#ifndef SLICE3D_MESH_H
#define SLICE3D_MESH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* MISE(resolution_0, depth, threshold) (mise.pyx:33-60); NULL if the parameters are out of range */
void* s3d_mise_create(int resolution0, int depth, double threshold);
void s3d_mise_destroy(void* mise);
int s3d_mise_resolution(void* mise);                         /* resolution_0 << depth */
/* query() (mise.pyx:106-129): count, then the (n,3) int64 grid points without a value, in insertion order */
long s3d_mise_query_count(void* mise);
void s3d_mise_query(void* mise, int64_t* out_points);
/* update(points (n,3) int64, values (n) float64) (mise.pyx:62-104,182-232): 0, or -1 if a point is not in the grid */
int s3d_mise_update(void* mise, const int64_t* points, const double* values, long n);
/* to_dense() (mise.pyx:131-163): (r,r,r) float64, r = resolution + 1 */
void s3d_mise_to_dense(void* mise, double* out);
long s3d_mise_num_points(void* mise);

/* marching_cubes(volume (nx,ny,nz) float64 C order, isovalue): result handle, sizes, copy-out, free */
void* s3d_mc_run(const double* grid, int nx, int ny, int nz, double iso);
long s3d_mc_num_vertices(void* result);
long s3d_mc_num_triangles(void* result);
void s3d_mc_copy(void* result, double* vertices /* (V,3) */, int64_t* triangles /* (F,3) */);
void s3d_mc_free(void* result);

#ifdef __cplusplus
}
#endif
#endif /* SLICE3D_MESH_H */
