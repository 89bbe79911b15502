"""Does run-time compilation work on the GPU box?  Compiles a kernel with hiprtc (ctypes), loads it with hipModuleLoadData and
runs it; prints compile time and code size.  (De-risks the generated-code formulation: DESIGN.md, known headroom.)"""
import ctypes as C, time, sys
rtc = C.CDLL('libhiprtc.so')
hip = C.CDLL('libamdhip64.so')
src = b'''
extern "C" __global__ void axpy(double a, const double* x, double* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { double r; asm("v_add_f64 %0, %1, |%2|" : "=v"(r) : "v"(y[i]), "v"(a * x[i])); y[i] = r; }
}
'''
prog = C.c_void_p()
assert rtc.hiprtcCreateProgram(C.byref(prog), src, b'axpy.hip', 0, None, None) == 0
opts = (C.c_char_p * 3)(b'--offload-arch=gfx950', b'-O3', b'-ffp-contract=off')
t0 = time.time()
rc = rtc.hiprtcCompileProgram(prog, 3, opts)
dt = time.time() - t0
n = C.c_size_t()
rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
log = C.create_string_buffer(n.value + 1)
rtc.hiprtcGetProgramLog(prog, log)
print('compile rc', rc, 'in %.2f s' % dt, log.value.decode()[:300])
assert rc == 0
rtc.hiprtcGetCodeSize(prog, C.byref(n))
code = C.create_string_buffer(n.value)
rtc.hiprtcGetCode(prog, code)
print('code object bytes', n.value)
assert hip.hipInit(0) == 0
mod, fn = C.c_void_p(), C.c_void_p()
assert hip.hipModuleLoadData(C.byref(mod), code) == 0
assert hip.hipModuleGetFunction(C.byref(fn), mod, b'axpy') == 0
N = 1 << 20
dx, dy = C.c_void_p(), C.c_void_p()
hip.hipMalloc(C.byref(dx), N * 8); hip.hipMalloc(C.byref(dy), N * 8)
import numpy as np
x = np.arange(N, dtype=np.float64) - N / 2; y = np.ones(N)
hip.hipMemcpy(dx, x.ctypes.data_as(C.c_void_p), N * 8, 1); hip.hipMemcpy(dy, y.ctypes.data_as(C.c_void_p), N * 8, 1)
a, nn = C.c_double(2.0), C.c_int(N)
args = (C.c_void_p * 4)(C.cast(C.pointer(a), C.c_void_p), C.cast(C.pointer(dx), C.c_void_p), C.cast(C.pointer(dy), C.c_void_p), C.cast(C.pointer(nn), C.c_void_p))
rc = hip.hipModuleLaunchKernel(fn, N // 256, 1, 1, 256, 1, 1, 0, None, args, None)
hip.hipDeviceSynchronize()
out = np.empty(N); hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), dy, N * 8, 2)
print('launch rc', rc, 'result ok', bool(np.array_equal(out, 1.0 + np.abs(2.0 * x))))
