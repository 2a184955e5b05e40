// pin_probe — what moving a one-shot batch (2^20 x 5 planes, 42 MB in, 8 MB out) over PCIe costs
// from PAGEABLE host memory, by method.  Decides how rafthip_batch_committed_index stages its
// buffers ("PCIe note").  Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/pin_probe tools/pin_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_touch(const unsigned long long* in, unsigned long long* out, size_t n_in, size_t n_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    unsigned long long a = 0;
    for (size_t s = 0; s < n_in / n_out; ++s) a ^= in[s * n_out + i];
    out[i] = a;
}

// two groups per lane, 16-byte loads
__global__ void k_touch2(const ulonglong2* in, ulonglong2* out, size_t n_in2, size_t n_out2) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out2) return;
    ulonglong2 a = {0, 0};
    for (size_t s = 0; s < n_in2 / n_out2; ++s) {
        const ulonglong2 v = in[s * n_out2 + i];
        a.x ^= v.x;
        a.y ^= v.y;
    }
    out[i] = a;
}

static void par_copy(char* dst, const char* src, size_t n, int threads) {
    std::vector<std::thread> th;
    const size_t per = (n / threads + 4095) & ~(size_t)4095;
    for (int t = 0; t < threads; ++t) {
        const size_t lo = (size_t)t * per, hi = lo + per < n ? lo + per : n;
        if (lo < hi) th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto& x : th) x.join();
}

int main() {
    const size_t n_in = (size_t)5 << 20, n_out = (size_t)1 << 20, bin = n_in * 8, bout = n_out * 8;
    unsigned long long* h_in = (unsigned long long*)aligned_alloc(4096, bin);
    unsigned long long* h_out = (unsigned long long*)aligned_alloc(4096, bout);
    for (size_t i = 0; i < n_in; ++i) h_in[i] = i * 0x9E3779B97F4A7C15ull;
    memset(h_out, 0, bout);
    unsigned long long *d_in, *d_out, *p_in, *p_out;
    CK(hipMalloc(&d_in, bin));
    CK(hipMalloc(&d_out, bout));
    CK(hipHostMalloc(&p_in, bin, hipHostMallocDefault));
    CK(hipHostMalloc(&p_out, bout, hipHostMallocDefault));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int R = 10;
    // (a) runtime pageable copies
    for (int w = 0; w < 2; ++w) { CK(hipMemcpy(d_in, h_in, bin, hipMemcpyHostToDevice)); CK(hipMemcpy(h_out, d_out, bout, hipMemcpyDeviceToHost)); }
    double t = now();
    for (int r = 0; r < R; ++r) CK(hipMemcpy(d_in, h_in, bin, hipMemcpyHostToDevice));
    double a = (now() - t) / R;
    t = now();
    for (int r = 0; r < R; ++r) CK(hipMemcpy(h_out, d_out, bout, hipMemcpyDeviceToHost));
    double b = (now() - t) / R;
    printf("pageable hipMemcpy   H2D %.3f ms (%.1f GB/s)   D2H %.3f ms (%.1f GB/s)\n", a * 1e3, bin / a / 1e9, b * 1e3, bout / b / 1e9);
    // (b) pinned copies
    t = now();
    for (int r = 0; r < R; ++r) CK(hipMemcpy(d_in, p_in, bin, hipMemcpyHostToDevice));
    a = (now() - t) / R;
    t = now();
    for (int r = 0; r < R; ++r) CK(hipMemcpy(p_out, d_out, bout, hipMemcpyDeviceToHost));
    b = (now() - t) / R;
    printf("pinned   hipMemcpy   H2D %.3f ms (%.1f GB/s)   D2H %.3f ms (%.1f GB/s)\n", a * 1e3, bin / a / 1e9, b * 1e3, bout / b / 1e9);
    // (c) register in place + unregister, a fresh buffer each time
    {
        double reg = 0, unreg = 0, kern = 0;
        int ok = 0;
        for (int r = 0; r < R; ++r) {
            unsigned long long* f_in = (unsigned long long*)aligned_alloc(4096, bin);
            unsigned long long* f_out = (unsigned long long*)aligned_alloc(4096, bout);
            memcpy(f_in, h_in, bin);
            memset(f_out, 0, bout);
            t = now();
            hipError_t e1 = hipHostRegister(f_in, bin, hipHostRegisterDefault);
            hipError_t e2 = hipHostRegister(f_out, bout, hipHostRegisterDefault);
            reg += now() - t;
            if (e1 != hipSuccess || e2 != hipSuccess) {
                printf("hipHostRegister: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
                (void)hipGetLastError();
                break;
            }
            void *zi, *zo;
            CK(hipHostGetDevicePointer(&zi, f_in, 0));
            CK(hipHostGetDevicePointer(&zo, f_out, 0));
            t = now();
            hipLaunchKernelGGL(k_touch, dim3(n_out / 256), dim3(256), 0, st, (unsigned long long*)zi, (unsigned long long*)zo, n_in, n_out);
            CK(hipStreamSynchronize(st));
            kern += now() - t;
            t = now();
            CK(hipHostUnregister(f_in));
            CK(hipHostUnregister(f_out));
            unreg += now() - t;
            free(f_in);
            free(f_out);
            ++ok;
        }
        if (ok) printf("register in place (40+8 MB): register %.3f ms, zero-copy kernel %.3f ms (%.1f GB/s in+out), unregister %.3f ms\n",
                       reg / ok * 1e3, kern / ok * 1e3, (bin + bout) / (kern / ok) / 1e9, unreg / ok * 1e3);
    }
    // (e) parallel CPU memcpy pageable -> pinned
    for (int threads : {1, 2, 4, 8, 12}) {
        par_copy((char*)p_in, (const char*)h_in, bin, threads);
        t = now();
        for (int r = 0; r < R; ++r) par_copy((char*)p_in, (const char*)h_in, bin, threads);
        a = (now() - t) / R;
        printf("CPU memcpy pageable->pinned, %2d threads (spawned per call): 40 MB %.3f ms (%.1f GB/s)\n", threads, a * 1e3, bin / a / 1e9);
    }
    // (f) zero-copy kernel over hipHostMalloc memory
    void *pi, *po;
    CK(hipHostGetDevicePointer(&pi, p_in, 0));
    CK(hipHostGetDevicePointer(&po, p_out, 0));
    for (int w = 0; w < 2; ++w) { hipLaunchKernelGGL(k_touch, dim3(n_out / 256), dim3(256), 0, st, (unsigned long long*)pi, (unsigned long long*)po, n_in, n_out); CK(hipStreamSynchronize(st)); }
    t = now();
    for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(k_touch, dim3(n_out / 256), dim3(256), 0, st, (unsigned long long*)pi, (unsigned long long*)po, n_in, n_out); CK(hipStreamSynchronize(st)); }
    a = (now() - t) / R;
    printf("zero-copy kernel over hipHostMalloc memory: %.3f ms (%.1f GB/s in+out)\n", a * 1e3, (bin + bout) / a / 1e9);
    // (g) the same kernel reading host memory but writing device memory: the read side of the link alone
    t = now();
    for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(k_touch, dim3(n_out / 256), dim3(256), 0, st, (unsigned long long*)pi, d_out, n_in, n_out); CK(hipStreamSynchronize(st)); }
    a = (now() - t) / R;
    printf("zero-copy reads, result to device memory:   %.3f ms (%.1f GB/s in)\n", a * 1e3, bin / a / 1e9);
    t = now();
    for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(k_touch2, dim3(n_out / 512), dim3(256), 0, st, (const ulonglong2*)pi, (ulonglong2*)d_out, n_in / 2, n_out / 2); CK(hipStreamSynchronize(st)); }
    a = (now() - t) / R;
    printf("zero-copy reads with 16-byte loads:         %.3f ms (%.1f GB/s in)\n", a * 1e3, bin / a / 1e9);
    t = now();
    for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(k_touch2, dim3(n_out / 512), dim3(256), 0, st, (const ulonglong2*)pi, (ulonglong2*)po, n_in / 2, n_out / 2); CK(hipStreamSynchronize(st)); }
    a = (now() - t) / R;
    printf("zero-copy 16-byte loads, result to host:    %.3f ms (%.1f GB/s in+out)\n", a * 1e3, (bin + bout) / a / 1e9);
    // (h) chunked DMA pipeline: pinned -> device by the copy engine, kernel on device memory, result back by
    // the copy engine on a second stream
    hipStream_t st2;
    CK(hipStreamCreate(&st2));
    for (size_t chunk : {(size_t)32768, (size_t)65536, (size_t)131072, (size_t)262144, (size_t)1048576}) {
        const size_t nchunks = n_out / chunk;
        std::vector<hipEvent_t> ev(nchunks);
        for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        double best = 1e9;
        for (int r = 0; r < R; ++r) {
            t = now();
            for (size_t k = 0; k < nchunks; ++k) {
                // pinned layout [chunk k][5][chunk]: one contiguous copy per chunk
                CK(hipMemcpyAsync(d_in + k * 5 * chunk, p_in + k * 5 * chunk, 5 * chunk * 8, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_touch, dim3(chunk / 256), dim3(256), 0, st, d_in + k * 5 * chunk, d_out + k * chunk, 5 * chunk, chunk);
                CK(hipEventRecord(ev[k], st));
                CK(hipStreamWaitEvent(st2, ev[k], 0));
                CK(hipMemcpyAsync(p_out + k * chunk, d_out + k * chunk, chunk * 8, hipMemcpyDeviceToHost, st2));
            }
            CK(hipStreamSynchronize(st));
            CK(hipStreamSynchronize(st2));
            const double d = now() - t;
            if (d < best) best = d;
        }
        printf("DMA pipeline, chunks of %7zu groups (%5.1f MB in): %.3f ms (%.1f GB/s in+out)\n", chunk, 5 * chunk * 8 / 1e6, best * 1e3, (bin + bout) / best / 1e9);
    }
    return 0;
}
