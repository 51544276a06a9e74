// verify.hip.h -- the second run of the window kernel (gfx950, wave = 64 lanes).
//
// Every busy window's rows are the reference's own arithmetic before window_kernel reads them (presence_kernel ->
// exact_rows_kernel, exact.hip.h).  What is left for a second run: a classic hit of the first run that stands on rows
// exact.hip.h has NOT covered -- a packet under the presence threshold, a hit born from noise.  Such a window is listed as a
// VerifyTask, its rows are marked (bm2) and recomputed by a second launch of exact_rows_kernel, and
//   verify_fill_kernel         copies the window's rows out of the (now exact) stream as the task's column of the time-major
//                              task stream dxt[pseudo-slot][kVerRows][drow]
//   window_kernel<LAY, true>   squelch figure carried over, clock recovery + slicer + access-code / LE search on that stream:
//                              the window's remaining records (kernels.hip.h)
// and finish_kernel continues them as before.
#pragma once
#include "kernels.hip.h"

namespace btgpu {

// The task stream the second run reads: dxt[(pseudo-slot * kVerRows + row) * drow + column], task q in column q % nch of
// pseudo-slot q / nch: the window's rows from the stream -- the 100-bin bank's tile-blocked copy dcol[tile][80][25] where it
// exists, else the strided column of d (row 0 is zeroed by the consumer, policy Q1).  One workgroup = 64 rows of one pseudo-slot;
// the values cross an LDS tile and leave as whole rows.
struct VerifyFillParams {
    const VerifyTask *tasks; const unsigned int *vcount; int vcap;
    const float *d; const float *dcol; int drow; long long d_rows;
    int nch, outs_per_slot, rows;     // rows of a task that are filled (min(ddc_out, kVerRows))
    float *dxt;
};
__global__ __launch_bounds__(256) void verify_fill_kernel(VerifyFillParams p)
{
    __shared__ float tile[64 * 81];
    unsigned int ntask = p.vcount[0];
    if (ntask > (unsigned int)p.vcap) ntask = (unsigned int)p.vcap;
    const int nps = (int)((ntask + (unsigned int)p.nch - 1u) / (unsigned int)p.nch);
    const int nrb = (p.rows + 63) / 64;
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
    for (int item = blockIdx.x; item < nps * nrb; item += gridDim.x) {
        const int ps = item / nrb, rb = item - ps * nrb;
        const int r = rb * 64 + lane;                              // row this lane fetches
        __syncthreads();
        for (int col = wv; col < p.nch; col += 4) {
            const unsigned int q = (unsigned int)(ps * p.nch + col);
            float v = 0.f;
            if (q < ntask && r < p.rows) {
                const VerifyTask tk = p.tasks[q];
                const int k = tk.w / p.nch, c = tk.w - k * p.nch;
                long long g = (long long)k * p.outs_per_slot + r;
                if (g >= p.d_rows) g = p.d_rows - 1;
                if (p.dcol) {
                    const unsigned int gq = (unsigned int)g, tq = gq / 25u;
                    v = p.dcol[(size_t)(gq + 25u * (79u * tq + (unsigned int)c))];
                } else v = p.d[(size_t)g * p.drow + c];
            }
            tile[lane * 81 + col] = v;
        }
        __syncthreads();
        const int nrows = p.rows - rb * 64 < 64 ? p.rows - rb * 64 : 64;
        for (int i = threadIdx.x; i < nrows * p.nch; i += 256) {
            const int rr = i / p.nch, cc = i - rr * p.nch;
            p.dxt[((size_t)ps * kVerRows + rb * 64 + rr) * p.drow + cc] = tile[rr * 81 + cc];
        }
    }
}

}  // namespace btgpu
