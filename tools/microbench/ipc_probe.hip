// Does HIP IPC map large allocations on this system?  ./ipc_probe.bin <GiB>   (parent allocates, child maps + reads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    double gib = argc > 1 ? atof(argv[1]) : 1.0;
    double imp_gib = argc > 2 ? atof(argv[2]) : 0.0; // device memory the importer holds in 1-GiB-or-less pieces before it maps
    int imp_pieces = argc > 3 ? atoi(argv[3]) : 1;
    size_t bytes = (size_t)(gib * (1ull << 30));
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 2;
    pid_t pid = fork(); // before any HIP call
    if (pid == 0)
    {
        hipIpcMemHandle_t h;
        if (read(p2c[0], &h, sizeof(h)) != (ssize_t)sizeof(h)) return 3;
        for (int i = 0; i < imp_pieces && imp_gib > 0; i++)
        {
            void *q = nullptr;
            hipError_t e2 = hipMalloc(&q, (size_t)(imp_gib / imp_pieces * (1ull << 30)));
            if (e2 != hipSuccess) printf("child: prealloc failed\n");
            hipMemset(q, 1, 4096);
        }
        hipDeviceSynchronize();
        double t0 = now();
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        printf("child: open -> %s in %.3f s\n", hipGetErrorString(e), now() - t0); fflush(stdout);
        if (e != hipSuccess) return 4;
        unsigned char *d = nullptr, back[256];
        hipMalloc((void **)&d, 256);
        t0 = now();
        e = hipMemcpy(d, (char *)p + bytes - 256, 256, hipMemcpyDeviceToDevice);
        hipMemcpy(back, d, 256, hipMemcpyDeviceToHost);
        printf("child: tail read -> %s in %.3f s, first byte %d\n", hipGetErrorString(e), now() - t0, (int)back[0]); fflush(stdout);
        hipIpcCloseMemHandle(p);
        char ok = 1;
        if (write(c2p[1], &ok, 1) != 1) return 5;
        return 0;
    }
    void *base = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc(&base, bytes);
    printf("parent: hipMalloc %.2f GiB -> %s\n", gib, hipGetErrorString(e)); fflush(stdout);
    hipMemset((char *)base + bytes - 256, 7, 256);
    hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    t0 = now();
    e = hipIpcGetMemHandle(&h, base);
    printf("parent: get handle -> %s in %.3f s\n", hipGetErrorString(e), now() - t0); fflush(stdout);
    if (write(p2c[1], &h, sizeof(h)) != (ssize_t)sizeof(h)) return 6;
    char ok = 0;
    if (read(c2p[0], &ok, 1) != 1) printf("parent: child did not answer\n");
    int st = 0;
    waitpid(pid, &st, 0);
    hipFree(base);
    return 0;
}
