// STFT -> |.|/norm -> sparse mel -> clamp -> 20 log10 -> affine, fused (reference: utils/audio.py:24-54 Spectrogram,
// :73-87 MelSpectrogram, data/abstractbasedataset.py:129-131 min-max normalisation).
//
// One 256-thread workgroup handles FT consecutive frames of one waveform:
//   * the (FT-1)*hop + 1024 samples those frames overlap on are read from HBM once, coalesced, into LDS
//     (the hop/n_fft = 1/4 overlap is served from LDS; out-of-range samples are the centre zero padding);
//   * frames are transformed two at a time (two real frames packed as one complex 1024-point signal, split
//     afterwards by conjugate symmetry) with a 5-pass radix-4 Stockham FFT in LDS, one butterfly per thread and
//     pass, twiddles from an LDS table;
//   * the mel projection uses the CSR form of the filterbank (<= 14 taps per row), results are collected in an
//     LDS [rows][FT] tile and written as FT-float row segments.
#include "pgv_common.h"

namespace {

constexpr int NFFT = 1024;
constexpr int NBIN = NFFT / 2 + 1;
constexpr int FT = 16;  // frames per workgroup

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__global__ __launch_bounds__(256) void stft_mel_kernel(
    const float* __restrict__ wav, int64_t n_samples, int hop, int n_frames, const float* __restrict__ window,
    float inv_norm, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, int n_rows, int use_mel, float floor_lin, float aff_a, float aff_b,
    float* __restrict__ out, int sig_len) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* tw = reinterpret_cast<float2*>(lds);            // [1024]
  float2* bufA = tw + NFFT;                               // [1024]
  float2* bufB = bufA + NFFT;                             // [1024]
  float* mag0 = reinterpret_cast<float*>(bufB + NFFT);    // [513]
  float* mag1 = mag0 + NBIN + 3;                          // [513]
  float* win = mag1 + NBIN + 3;                           // [1024]
  float* tile = win + NFFT;                               // [n_rows][FT+1]
  float* sig = tile + (size_t)n_rows * (FT + 1);          // [sig_len]

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * FT;
  const int nf = min(FT, n_frames - f0);
  const float* w = wav + (int64_t)b * n_samples;

  for (int i = tid; i < NFFT; i += 256) {
    float s, c;
    sincospif(-2.0f * (float)i / (float)NFFT, &s, &c);
    tw[i] = make_float2(c, s);
    win[i] = window[i];
  }
  const int64_t s0 = (int64_t)f0 * hop - NFFT / 2;
  for (int i = tid; i < sig_len; i += 256) {
    const int64_t g = s0 + i;
    sig[i] = (g >= 0 && g < n_samples) ? w[g] : 0.f;
  }
  __syncthreads();

  for (int fp = 0; fp < nf; fp += 2) {
    const bool has2 = fp + 1 < nf;
    // load: z = frame(fp) + i*frame(fp+1), windowed
    for (int i = tid; i < NFFT; i += 256) {
      const float wv = win[i];
      const float re = sig[fp * hop + i] * wv;
      const float im = has2 ? sig[(fp + 1) * hop + i] * wv : 0.f;
      bufA[i] = make_float2(re, im);
    }
    __syncthreads();
    float2* src = bufA;
    float2* dst = bufB;
#pragma unroll
    for (int pass = 0; pass < 5; ++pass) {
      const int Ns = 1 << (2 * pass);
      const int j = tid;
      const int k = j & (Ns - 1);
      const int tws = (NFFT / 4) / Ns;  // twiddle index step: 1024/(4*Ns)
      float2 v0 = src[j], v1 = src[j + 256], v2 = src[j + 512], v3 = src[j + 768];
      if (pass > 0) {
        v1 = cmul(v1, tw[k * tws]);
        v2 = cmul(v2, tw[2 * k * tws]);
        v3 = cmul(v3, tw[3 * k * tws]);
      }
      // radix-4 butterfly (forward transform: -i rotation)
      const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y), a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
      const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y), a3 = make_float2(v1.x - v3.x, v1.y - v3.y);
      const float2 o0 = make_float2(a0.x + a2.x, a0.y + a2.y);
      const float2 o2 = make_float2(a0.x - a2.x, a0.y - a2.y);
      const float2 o1 = make_float2(a1.x + a3.y, a1.y - a3.x);  // a1 - i*a3
      const float2 o3 = make_float2(a1.x - a3.y, a1.y + a3.x);  // a1 + i*a3
      const int j0 = ((j - k) << 2) + k;  // (j/Ns)*Ns*4 + k
      dst[j0] = o0;
      dst[j0 + Ns] = o1;
      dst[j0 + 2 * Ns] = o2;
      dst[j0 + 3 * Ns] = o3;
      __syncthreads();
      float2* t = src;
      src = dst;
      dst = t;
    }
    // src holds Z. X1[k] = (Z[k] + conj(Z[N-k]))/2 ; X2[k] = (Z[k] - conj(Z[N-k]))/(2i)
    for (int k = tid; k < NBIN; k += 256) {
      const float2 zk = src[k];
      const float2 zn = src[(NFFT - k) & (NFFT - 1)];
      const float x1r = 0.5f * (zk.x + zn.x), x1i = 0.5f * (zk.y - zn.y);
      const float x2r = 0.5f * (zk.y + zn.y), x2i = -0.5f * (zk.x - zn.x);
      mag0[k] = sqrtf(x1r * x1r + x1i * x1i) * inv_norm;
      mag1[k] = sqrtf(x2r * x2r + x2i * x2i) * inv_norm;
    }
    __syncthreads();
    for (int r = tid; r < n_rows; r += 256) {
      float m0, m1;
      if (use_mel) {
        m0 = 0.f;
        m1 = 0.f;
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) {
          const int c = col[e];
          const float v = val[e];
          m0 = fmaf(v, mag0[c], m0);
          m1 = fmaf(v, mag1[c], m1);
        }
      } else {
        m0 = mag0[r];
        m1 = mag1[r];
      }
      tile[r * (FT + 1) + fp] = fmaf(aff_a, 20.0f * log10f(fmaxf(m0, floor_lin)), aff_b);
      if (has2) tile[r * (FT + 1) + fp + 1] = fmaf(aff_a, 20.0f * log10f(fmaxf(m1, floor_lin)), aff_b);
    }
    __syncthreads();
  }
  // write the [n_rows][nf] tile: lanes run along frames inside a row segment
  float* o = out + (int64_t)b * n_rows * n_frames;
  for (int i = tid; i < n_rows * FT; i += 256) {
    const int r = i / FT, f = i % FT;
    if (f < nf) o[(int64_t)r * n_frames + f0 + f] = tile[r * (FT + 1) + f];
  }
}

}  // namespace

extern "C" int pgv_stft_mel(const float* wav, int B, int64_t n_samples, int n_fft, int hop, int n_frames,
                            const float* window, float norm, const int32_t* mel_row_ptr, const int32_t* mel_col,
                            const float* mel_val, int n_mels, float floor_lin, float affine_a, float affine_b,
                            float* out, void* stream) {
  PGV_CHECK_ARG(n_fft == NFFT, "pgv_stft_mel: only n_fft=1024 is implemented (got %d)", n_fft);
  PGV_CHECK_ARG(wav && window && out && B >= 0 && n_samples >= 0 && hop > 0 && hop <= NFFT && n_frames > 0 &&
                    norm > 0.f,
                "pgv_stft_mel: bad argument");
  PGV_CHECK_ARG(n_mels == 0 || (mel_row_ptr && mel_col && mel_val), "pgv_stft_mel: mel CSR missing");
  PGV_CHECK_ARG((int64_t)(n_frames - 1) * hop <= n_samples,
                "pgv_stft_mel: n_frames=%d exceeds 1 + n_samples/hop (centre padding)", n_frames);
  if (B == 0) return PGV_OK;
  const int n_rows = n_mels > 0 ? n_mels : NBIN;
  const int sig_len = (FT - 1) * hop + NFFT;
  const size_t lds_bytes = sizeof(float2) * 3 * NFFT + sizeof(float) * (2 * (NBIN + 3) + NFFT) +
                           sizeof(float) * ((size_t)n_rows * (FT + 1) + sig_len);
  PGV_CHECK_ARG(lds_bytes <= 160 * 1024, "pgv_stft_mel: %d output rows need %zu B of LDS", n_rows, lds_bytes);
  static bool attr_set = false;  // idempotent; only widens the dynamic-LDS cap of this kernel
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)stft_mel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid((unsigned)pgv_cdiv(n_frames, FT), (unsigned)B);
  hipLaunchKernelGGL(stft_mel_kernel, grid, dim3(256), lds_bytes, pgv_stream(stream), wav, n_samples, hop, n_frames,
                     window, 1.0f / norm, mel_row_ptr, mel_col, mel_val, n_rows, n_mels > 0 ? 1 : 0, floor_lin,
                     affine_a, affine_b, out, sig_len);
  PGV_CHECK_LAUNCH("stft_mel");
  return PGV_OK;
}
