// wav -> log-mel front-end of the reference (meldataset.py:73-96) on the GPU, fp32 throughout:
//   reflect-pad (n_fft - hop)/2, frame + periodic Hann window     -> frames [B*F, n_fft]          (frame_window_kernel)
//   real DFT as ONE fp32-MFMA GEMM against a [cos | -sin] basis    -> [B*F, 2*(n_fft/2+1)]         (gemm_nt_kernel<float>)
//   magnitude sqrt(re^2 + im^2 + 1e-9)                             -> [B*F, bins]                  (mag_kernel)
//   mel filterbank as a second GEMM, log(clamp(., 1e-5))           -> [B, n_mels, F] channels-first (logmel_kernel)
// n_fft = 1024 is a dense 1024 x 1026 contraction per frame: at the frame counts of an inference batch (thousands of rows)
// that is a few GFLOP on the matrix cores and needs no FFT library; the output lands directly in the layout the encoder reads.
#pragma once
#include "common.cuh"

namespace ge2e {

// one block per frame row
__global__ void __launch_bounds__(256) frame_window_kernel(const float* y, float* A, int L, int F, int n_fft, int hop, int pad) {
    const int row = blockIdx.x, b = row / F, f = row % F;
    for (int k = threadIdx.x; k < n_fft; k += 256) {
        int src = f * hop + k - pad;
        if (src < 0) src = -src;                              // reflect without repeating the edge sample (torch 'reflect')
        if (src >= L) src = 2 * (L - 1) - src;
        src = src < 0 ? 0 : src;
        const float w = 0.5f - 0.5f * cospif(2.0f * (float)k / (float)n_fft);   // torch.hann_window(periodic=True)
        A[(size_t)row * n_fft + k] = y[(size_t)b * L + src] * w;
    }
}

// W[r][n]: r < bins: cos(2 pi r n / N); bins <= r < 2 bins: -sin(2 pi (r - bins) n / N); beyond: 0.  The phase index is
// reduced modulo N in integers first, so the table is exact to double rounding before the cast.
__global__ void __launch_bounds__(256) dft_basis_kernel(float* W, int n_fft, int bins, int rows) {
    const int r = blockIdx.x;
    for (int n = threadIdx.x; n < n_fft; n += 256) {
        float v = 0.0f;
        if (r < 2 * bins) {
            const int k = r < bins ? r : r - bins;
            const int m = (int)(((long long)k * n) % n_fft);
            const double a = 2.0 * (double)m / (double)n_fft;
            v = (float)(r < bins ? cospi(a) : -sinpi(a));
        }
        W[(size_t)r * n_fft + n] = v;
    }
}

__global__ void __launch_bounds__(256) mag_kernel(const float* S, int lds, float* Mg, int ldm, int bins, int rows) {
    const int row = blockIdx.x;
    for (int k = threadIdx.x; k < ldm; k += 256) {
        float v = 0.0f;
        if (k < bins) { const float re = S[(size_t)row * lds + k], im = S[(size_t)row * lds + bins + k]; v = sqrtf(re * re + im * im + 1e-9f); }
        Mg[(size_t)row * ldm + k] = v;
    }
}

__global__ void __launch_bounds__(256) pad_basis_kernel(const float* basis, float* P, int n_mels, int bins, int ldp) {
    const int m = blockIdx.x;                                 // 128 rows
    for (int k = threadIdx.x; k < ldp; k += 256) P[(size_t)m * ldp + k] = (m < n_mels && k < bins) ? basis[(size_t)m * bins + k] : 0.0f;
}

// out[b][m][f] = log(max(mel[(b*F + f)][m], 1e-5)); grid = (ceil(F / 32), B), a 32 x 32 LDS transpose per mel block
__global__ void __launch_bounds__(256) logmel_kernel(const float* Ml, float* out, int F, int n_mels) {
    __shared__ float tile[32][33];
    const int f0 = blockIdx.x * 32, b = blockIdx.y;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int m0 = 0; m0 < n_mels; m0 += 32) {
        for (int r = ly; r < 32; r += 8) {
            const int f = f0 + r, m = m0 + lx;
            tile[r][lx] = (f < F && m < n_mels) ? Ml[((size_t)b * F + f) * 128 + m] : 1.0f;
        }
        __syncthreads();
        for (int r = ly; r < 32; r += 8) {
            const int m = m0 + r, f = f0 + lx;
            if (m < n_mels && f < F) out[((size_t)b * n_mels + m) * F + f] = logf(fmaxf(tile[lx][r], 1e-5f));
        }
        __syncthreads();
    }
}

}  // namespace ge2e
