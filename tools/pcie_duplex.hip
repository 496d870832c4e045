// Is the host link full duplex here?  Pinned 256 MB buffers (first-touched on the cpus next to the device),
// hipMemcpyAsync H2D only / D2H only / both at once on two streams -- no helper threads, no kernels.
// Also: the same with the transfers cut into 6 MB pieces (the slab pipeline's copy size), and with both
// directions issued on ONE stream.  GB/s per direction.
// Build: hipcc --offload-arch=gfx950 -O3 tools/pcie_duplex.hip -o tools/pcie_duplex
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <sched.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int numa_node_of_device(int dev) {
  char bus[64];
  if (hipDeviceGetPCIBusId(bus, sizeof bus, dev) != hipSuccess) return -1;
  for (char* p = bus; *p; ++p) if (*p >= 'A' && *p <= 'F') *p += 32;
  char path[256];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  int n = -1;
  if (fscanf(f, "%d", &n) != 1) n = -1;
  fclose(f);
  return n;
}
static void bind_to_node(int node) {
  if (node < 0) return;
  char path[128];
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) return;
  cpu_set_t set; CPU_ZERO(&set);
  int a, b; char c;
  while (fscanf(f, "%d", &a) == 1) {
    b = a;
    if (fscanf(f, "%c", &c) == 1 && c == '-') { if (fscanf(f, "%d", &b) != 1) b = a; if (fscanf(f, "%c", &c) != 1) c = 0; }
    for (int i = a; i <= b; ++i) CPU_SET(i, &set);
    if (c != ',') break;
  }
  fclose(f);
  sched_setaffinity(0, sizeof set, &set);
}

int main() {
  const size_t N = 256u << 20;
  const int node = numa_node_of_device(0);
  bind_to_node(node);
  CHECK(hipSetDevice(0));
  void *h_in, *h_out, *d_in, *d_out;
  CHECK(hipHostMalloc(&h_in, N, hipHostMallocDefault));
  CHECK(hipHostMalloc(&h_out, N, hipHostMallocDefault));
  memset(h_in, 1, N); memset(h_out, 2, N);
  CHECK(hipMalloc(&d_in, N)); CHECK(hipMalloc(&d_out, N));
  CHECK(hipMemset(d_out, 3, N));
  hipStream_t s0, s1;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  printf("device 0 on NUMA node %d; pinned buffers of %zu MB\n", node, N >> 20);
  const int reps = 8;
  auto run = [&](const char* name, bool up, bool down, size_t piece, bool one_stream) -> int {
    for (int warm = 0; warm < 2; ++warm) {
      const double t0 = now();
      for (int r = 0; r < reps; ++r)
        for (size_t off = 0; off < N; off += piece) {
          const size_t n = off + piece <= N ? piece : N - off;
          if (up) CHECK(hipMemcpyAsync((char*)d_in + off, (char*)h_in + off, n, hipMemcpyHostToDevice, s0));
          if (down) CHECK(hipMemcpyAsync((char*)h_out + off, (char*)d_out + off, n, hipMemcpyDeviceToHost, one_stream ? s0 : s1));
        }
      CHECK(hipStreamSynchronize(s0));
      CHECK(hipStreamSynchronize(s1));
      const double dt = now() - t0;
      if (warm) {
        const double gbs = (double)N * reps / dt / 1e9;
        printf("%-58s %6.1f ms per 256 MB", name, dt / reps * 1e3);
        if (up) printf("   H2D %5.1f GB/s", gbs);
        if (down) printf("   D2H %5.1f GB/s", gbs);
        if (up && down) printf("   sum %5.1f GB/s", 2 * gbs);
        printf("\n");
      }
    }
    return 0;
  };
  if (run("H2D only, one 256 MB copy", true, false, N, false)) return 1;
  if (run("D2H only, one 256 MB copy", false, true, N, false)) return 1;
  if (run("both at once, two streams, 256 MB copies", true, true, N, false)) return 1;
  if (run("H2D only, 6 MB pieces", true, false, 6u << 20, false)) return 1;
  if (run("D2H only, 6 MB pieces", false, true, 6u << 20, false)) return 1;
  if (run("both at once, two streams, 6 MB pieces", true, true, 6u << 20, false)) return 1;
  if (run("both, ONE stream, 6 MB pieces alternating", true, true, 6u << 20, true)) return 1;
  if (run("both at once, two streams, 1 MB pieces", true, true, 1u << 20, false)) return 1;
  return 0;
}
