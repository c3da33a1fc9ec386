// TEST INFRASTRUCTURE ONLY — thin C driver around the REFERENCE's marching-cubes template
// (/root/reference/reg_slices/src_convonet/utils/libmcubes/marchingcubes.h, compiled where it lies by
// oracle/Makefile into oracle/_ref/libmcref.so).  Mirrors libmcubes/pywrapper.cpp:90-127 without NumPy.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "marchingcubes.h"

struct ArrFn {
    const double* a; long ny, nz;
    double operator()(double x, double y, double z) const { return a[((long)x * ny + (long)y) * nz + (long)z]; }
};
static std::vector<double> g_v;
static std::vector<size_t> g_p;
extern "C" long mcref_run(const double* grid, int nx, int ny, int nz, double iso, long* ntri_idx) {
    g_v.clear(); g_p.clear();
    double lower[3] = {0, 0, 0}, upper[3] = {(double)nx - 1, (double)ny - 1, (double)nz - 1};
    mc::marching_cubes<double>(lower, upper, nx, ny, nz, ArrFn{grid, ny, nz}, iso, g_v, g_p);
    *ntri_idx = (long)g_p.size();
    return (long)g_v.size();
}
extern "C" void mcref_copy(double* v, int64_t* p) {
    memcpy(v, g_v.data(), g_v.size() * sizeof(double));
    for (size_t i = 0; i < g_p.size(); ++i) p[i] = (int64_t)g_p[i];
}
