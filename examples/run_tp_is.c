/* A complete non-Python host of the hot path: one MessagePackBlock launch (hg_tp_is) from a table file.
 *
 *   build (here, cross-compiles):  hipcc --offload-arch=gfx950 -x hip -I include examples/run_tp_is.c -L hamgnn_amd/lib -lhamgnn_hip -Wl,-rpath,'$ORIGIN/../hamgnn_amd/lib' -o examples/run_tp_is
 *   run (GPU box):                 examples/run_tp_is block.hgprog inputs.bin expected.bin
 *
 * block.hgprog : written by hamgnn_amd/export.py (tables + packed weights; include/hamgnn_tables.h)
 * inputs.bin   : int64 rows, int64 nsrc, int64 src_dim, int64 hid_stride; then float32 src[nsrc][rows][src_dim] (edge-frame planar rows), float32 h_node[rows][hid_stride],
 *                float32 h_edge[rows][hid_stride] (the radial MLPs' hidden activations)
 * expected.bin : float32 out[rows][out_dim] from the Python host (ops.tp_fused on the same tables)
 * Prints the max |difference| and exits 0 when it is below 1e-6 * max |expected| (same kernel, same tables: bit-equal in single-part launches).
 * Nothing here is Python, torch or the planner: only the C ABI of include/hamgnn_hip.h and the HIP runtime for device memory.                                        */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include "hamgnn_hip.h"
#include "hamgnn_tables.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static void* upload(const void* host, size_t nbytes) {
    void* d = NULL;
    if (hipMalloc(&d, nbytes ? nbytes : 4) != hipSuccess) return NULL;
    if (nbytes && hipMemcpy(d, host, nbytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s block.hgprog inputs.bin expected.bin\n", argv[0]); return 1; }
    HgProgFile P;
    int rc = hg_prog_load(argv[1], &P);
    if (rc) { fprintf(stderr, "cannot load %s (%d)\n", argv[1], rc); return 1; }
    FILE* fi = fopen(argv[2], "rb");
    if (!fi) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
    int64_t hd[4];
    if (fread(hd, 8, 4, fi) != 4) return 1;
    const int64_t rows = hd[0], nsrc = hd[1], sdim = hd[2], hstride = hd[3];
    const size_t src_bytes = (size_t)rows * sdim * 4, hid_bytes = (size_t)rows * hstride * 4, out_bytes = (size_t)rows * P.out_dim * 4;
    float* hbuf = (float*)malloc(src_bytes > hid_bytes ? src_bytes : hid_bytes);
    const float* d_src[4] = {0, 0, 0, 0};
    int64_t strides[4] = {sdim, sdim, sdim, sdim};
    for (int s = 0; s < nsrc; ++s) {
        if (fread(hbuf, 1, src_bytes, fi) != src_bytes) return 1;
        d_src[s] = (const float*)upload(hbuf, src_bytes);
    }
    if (fread(hbuf, 1, hid_bytes, fi) != hid_bytes) return 1;
    const float* d_hn = (const float*)upload(hbuf, hid_bytes);
    if (fread(hbuf, 1, hid_bytes, fi) != hid_bytes) return 1;
    const float* d_he = (const float*)upload(hbuf, hid_bytes);
    fclose(fi);
    const float* d_w = (const float*)upload(P.weights.data, (size_t)P.weights.nbytes);
    const int32_t* d_seg = (const int32_t*)upload(P.seg_table.data, (size_t)P.seg_table.nbytes);
    const int32_t* d_blk = (const int32_t*)upload(P.block_table.data, (size_t)P.block_table.nbytes);
    const int32_t* d_ph = (const int32_t*)upload(P.phase_table.data, (size_t)P.phase_table.nbytes);
    const int32_t* d_grp = (const int32_t*)upload(P.group_table.data, (size_t)P.group_table.nbytes);
    const int32_t* d_it = (const int32_t*)upload(P.item_table.data, (size_t)P.item_table.nbytes);
    const int32_t* d_part = (const int32_t*)upload(P.part_table.data, (size_t)P.part_table.nbytes);
    const int32_t* d_row = (const int32_t*)upload(P.row_table.data, (size_t)P.row_table.nbytes);
    float* d_out = NULL;
    CK(hipMalloc((void**)&d_out, out_bytes));
    if (P.zero_fill_out) CK(hipMemset(d_out, 0, out_bytes));   /* parts that share segments ADD their tiles (include/hamgnn_hip.h) */
    if (hstride < P.hidden) { fprintf(stderr, "hidden rows narrower than the program's padded hidden width\n"); return 1; }
    rc = hg_tp_is(d_src, strides, (int)nsrc, d_hn, d_he, P.hidden, NULL, 0, NULL, d_w, d_seg, d_blk, d_ph, d_grp, d_it, d_part, (const int32_t*)P.part_table.data, P.nparts,
                  d_row, P.lds_bytes, NULL, 0, NULL, NULL, d_out, (int64_t)P.out_dim, rows, NULL);
    if (rc) { fprintf(stderr, "hg_tp_is failed: %d (%s)\n", rc, hg_last_error()); return 3; }
    CK(hipDeviceSynchronize());
    float* out = (float*)malloc(out_bytes), *want = (float*)malloc(out_bytes);
    CK(hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost));
    FILE* fe = fopen(argv[3], "rb");
    if (!fe || fread(want, 1, out_bytes, fe) != out_bytes) { fprintf(stderr, "cannot read %s\n", argv[3]); return 1; }
    fclose(fe);
    double dmax = 0.0, wmax = 0.0;
    for (size_t i = 0; i < out_bytes / 4; ++i) {
        const double d = fabs((double)out[i] - (double)want[i]);
        if (d > dmax) dmax = d;
        if (fabs((double)want[i]) > wmax) wmax = fabs((double)want[i]);
    }
    printf("C host: rows %lld, out_dim %d, nparts %d: max |C host - Python host| = %.3e (max |expected| %.3e)\n", (long long)rows, P.out_dim, P.nparts, dmax, wmax);
    hg_prog_free(&P);
    return dmax <= 1e-6 * wmax ? 0 : 4;
}
