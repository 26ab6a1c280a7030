"""What the matrix cores sustain on this chip with NO memory traffic: a register-only MFMA loop
(4 independent accumulators per wave, 2 waves per SIMD), for all-zero and for N(0,1)-like operands.
Calibrates how much of a GEMM's distance from the datasheet peak is the kernel and how much is the
clock the chip holds under that load.   usage: python tools/bench_mfma_peak.py"""
import ctypes as C, json, os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytensor_amd import ffi, kernel_cache
from pytensor_amd.device import DeviceArray
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import timed

SRC = r'''
typedef float f16v __attribute__((ext_vector_type(16)));
typedef double d4v __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(256) void mfma_f32_loop(float* out, long long iters, float seed) {
  f16v acc[4];
  for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
  float a = seed * (float)((threadIdx.x * 37 % 101) - 50) * 0.02f, b = seed * (float)((threadIdx.x * 53 % 97) - 48) * 0.02f;
  for (long long i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
    }
    a = -a; b = -b;  // keeps the accumulators bounded
  }
  float s = 0.f;
  for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) s += acc[q][r];
  out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" __global__ __launch_bounds__(256) void mfma_f64_loop(double* out, long long iters, double seed) {
  d4v acc[8];
  for (int q = 0; q < 8; q++) for (int r = 0; r < 4; r++) acc[q][r] = 0.0;
  double a = seed * (double)((threadIdx.x * 37 % 101) - 50) * 0.02, b = seed * (double)((threadIdx.x * 53 % 97) - 48) * 0.02;
  for (long long i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int q = 0; q < 8; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64((q & 1) ? a : b, (q & 2) ? a : b, acc[q], 0, 0, 0);
    }
    a = -a; b = -b;
  }
  double s = 0.0;
  for (int q = 0; q < 8; q++) for (int r = 0; r < 4; r++) s += acc[q][r];
  out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
}
'''

ffi.init(0)
lib = ffi.lib()
grid = 512  # 2 workgroups of 4 waves per CU
out = DeviceArray.empty((grid * 256,), "float64")
for name, flops_per_iter, pk, fmt in (("mfma_f32_loop", 32 * 2 * 32 * 32 * 2, 157.3, "<qqf4x"), ("mfma_f64_loop", 32 * 2 * 16 * 16 * 4, 78.6, "<qqd")):
    fn = kernel_cache.get_function(SRC, name)
    for seed in (0.0, 1.0):
        iters = 20000
        buf = struct.pack(fmt, out.ptr, iters, seed)
        run = lambda: ffi.check(lib.pthip_launch(fn, grid, 1, 1, 256, 1, 1, 0, buf, len(buf)))
        ms = timed(lib, run, 5)
        tf = flops_per_iter * iters * grid * 4 / ms / 1e9
        print(json.dumps({"kernel": name, "operands": "zero" if seed == 0 else "nonzero", "ms": round(ms, 3), "TFLOPs": round(tf, 1), "frac_of_datasheet": round(tf / pk, 3)}))
