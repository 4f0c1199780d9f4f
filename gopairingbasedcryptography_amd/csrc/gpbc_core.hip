// libgpbc_bn254.so, unit 1 of 4: process-wide state and lifetime entries of the C ABI (include/gpbc_bn254.h), the per-stream
// internal workspace, and the field-level test entry.  gfx950 only.
#include "gpbc_common.hpp"

thread_local char g_err[512] = "";
std::atomic<int> g_device{-1};
std::mutex g_ws_seq_mu;

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
int bind_device() {
    int d = g_device.load();
    if (d < 0) return fail(GPBC_ERR_NO_DEVICE, "gpbc_init() has not bound a HIP device (no CPU fallback exists)");
    HIP_TRY(hipSetDevice(d));
    return GPBC_OK;
}
int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GPBC_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return GPBC_OK;
}
int sync_default() { HIP_TRY(hipStreamSynchronize(nullptr)); return GPBC_OK; }

struct StreamWs { int device; hipStream_t stream; void *ptr; size_t bytes; };
static std::mutex g_ws_mu;
static std::vector<StreamWs> g_ws;
int stream_workspace(hipStream_t stream, size_t bytes, int32_t **out) {
    int dev = g_device.load();
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (auto &w : g_ws)
        if (w.device == dev && w.stream == stream) {
            if (w.bytes < bytes) {
                HIP_TRY(hipStreamSynchronize(stream));
                HIP_TRY(hipFree(w.ptr));
                w.ptr = nullptr; w.bytes = 0;
                HIP_TRY(hipMalloc(&w.ptr, bytes));
                w.bytes = bytes;
            }
            *out = (int32_t *)w.ptr;
            return GPBC_OK;
        }
    void *ptr = nullptr;
    HIP_TRY(hipMalloc(&ptr, bytes));
    g_ws.push_back(StreamWs{dev, stream, ptr, bytes});
    *out = (int32_t *)ptr;
    return GPBC_OK;
}
void free_workspaces() {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (auto &w : g_ws) if (w.ptr) { (void)hipSetDevice(w.device); (void)hipFree(w.ptr); }
    g_ws.clear();
}

__global__ void __launch_bounds__(BLOCK) k_fp_mul(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint8_t *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fe_store(out + i * 32, fe_mul(fe_load(a + i * 32), fe_load(b + i * 32)));
}

extern "C" {

int gpbc_abi_version(void) { return 4; }
const char *gpbc_last_error(void) { return g_err; }

int gpbc_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(GPBC_ERR_NO_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return n;
}

int gpbc_init(int device) {
    int n = gpbc_device_count();
    if (n <= 0) return fail(GPBC_ERR_NO_DEVICE, "no HIP device visible (this engine has no CPU fallback)");
    if (device < 0 || device >= n) return fail(GPBC_ERR_INVALID_ARG, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GPBC_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    g_device.store(device);
    return GPBC_OK;
}

int gpbc_shutdown(void) {
    free_workspaces();
    g_device.store(-1);
    return GPBC_OK;
}

int gpbc_fp_mul_batch(const void *a, const void *b, size_t n, void *out) {
    if (!n) return GPBC_OK;
    if (!a || !b || !out) return fail(GPBC_ERR_INVALID_ARG, "null pointer");
    TRY(bind_device());
    DevBuf dA, dB, dO;
    TRY(dA.upload(a, n * 32)); TRY(dB.upload(b, n * 32)); TRY(dO.alloc(n * 32));
    k_fp_mul<<<grid_for(n), BLOCK>>>(dA.u8(), dB.u8(), dO.u8(), n);
    TRY(check_launch("k_fp_mul"));
    TRY(sync_default());
    return dO.download(out, n * 32);
}

}  // extern "C"
