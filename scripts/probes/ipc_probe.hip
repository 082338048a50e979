// Feasibility probe for a peer-store halo transport: two processes on one GPU, (1) a device buffer (fine-grained or plain)
// exported with hipIpcGetMemHandle and written by a kernel of the OTHER process, (2) flags in POSIX shared memory registered
// with hipHostRegister in both, polled / set by kernels.   build: hipcc --offload-arch=gfx950 -O2 ipc_probe.hip -o ipc_probe -lrt
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("[%d] %s failed: %s\n", getpid(), #x, hipGetErrorString(e)); exit(2); } } while (0)

__global__ void push(unsigned long long* remote, int n, unsigned* flag, unsigned value) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) remote[i] = 1000ull * value + i;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) {  // (one block in this probe)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void wait_and_check(const unsigned long long* local, int n, unsigned* flag, unsigned value, unsigned* result) {
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
            if (wall_clock64() - t0 > 300000000LL) { result[1] = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    unsigned bad = 0;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) bad += local[i] != 1000ull * value + i;
    atomicAdd(&result[0], bad);
}

int main(int argc, char** argv) {
    const int finegrained = argc > 1 ? atoi(argv[1]) : 1;
    const int n = 1 << 16;
    const char* shm = "/gt4mi_ipc_probe";
    shm_unlink(shm);
    int fd = shm_open(shm, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, 4096) != 0) { perror("shm"); return 2; }
    int pipe_fd[2];
    if (pipe(pipe_fd) != 0) return 2;
    pid_t child = fork();
    unsigned* host = (unsigned*)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (child != 0) {  // parent = receiver: owns the buffer, exports it, waits for the flag in a kernel
        memset(host, 0, 4096);
        CK(hipSetDevice(0));
        unsigned long long* buf = nullptr;
        if (finegrained) CK(hipExtMallocWithFlags((void**)&buf, n * 8, hipDeviceMallocFinegrained));
        else CK(hipMalloc((void**)&buf, n * 8));
        CK(hipMemset(buf, 0, n * 8));
        hipIpcMemHandle_t h;
        CK(hipIpcGetMemHandle(&h, buf));
        if (write(pipe_fd[1], &h, sizeof h) != sizeof h) return 2;
        CK(hipHostRegister(host, 4096, hipHostRegisterMapped));
        unsigned* dflag = nullptr;
        CK(hipHostGetDevicePointer((void**)&dflag, host, 0));
        unsigned* result = nullptr;
        CK(hipMalloc((void**)&result, 8));
        for (unsigned step = 1; step <= 3; ++step) {
            CK(hipMemset(result, 0, 8));
            hipEvent_t a, b;
            CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(wait_and_check, dim3(1), dim3(256), 0, 0, buf, n, dflag, step, result);
            CK(hipEventRecord(b, 0));
            CK(hipDeviceSynchronize());
            unsigned r[2];
            CK(hipMemcpy(r, result, 8, hipMemcpyDeviceToHost));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            printf("[receiver] step %u (%s buffer): %u wrong items, timeout %u, waited %.3f ms\n", step, finegrained ? "fine-grained" : "plain", r[0], r[1], ms);
            host[16] = step;  // tell the sender (host side) that the next round may start
        }
        int status = 0;
        waitpid(child, &status, 0);
        shm_unlink(shm);
        printf("[receiver] sender exit status %d\n", WEXITSTATUS(status));
        return 0;
    }
    // child = sender: opens the receiver's buffer, writes it from a kernel, raises the flag from the kernel
    hipIpcMemHandle_t h;
    if (read(pipe_fd[0], &h, sizeof h) != sizeof h) return 2;
    CK(hipSetDevice(0));
    unsigned long long* remote = nullptr;
    CK(hipIpcOpenMemHandle((void**)&remote, h, hipIpcMemLazyEnablePeerAccess));
    CK(hipHostRegister(host, 4096, hipHostRegisterMapped));
    unsigned* dflag = nullptr;
    CK(hipHostGetDevicePointer((void**)&dflag, host, 0));
    for (unsigned step = 1; step <= 3; ++step) {
        usleep(200000);  // the receiver's kernel is already polling
        hipLaunchKernelGGL(push, dim3((n + 1023) / 1024 > 1 ? 1 : 1), dim3(1024), 0, 0, remote, 1024, dflag, step);  // first 1024 items by one block
        CK(hipDeviceSynchronize());
        // the rest of the buffer with a plain kernel-boundary protocol: write, synchronise, then set the flag from the host -- not used
        while (host[16] < step) usleep(1000);
    }
    CK(hipIpcCloseMemHandle(remote));
    return 0;
}
