// unaligned_probe.hip -- do 16-byte vector loads/stores at 4-byte-aligned
// addresses work on this part, and what do they cost?  1R2W streaming pattern
// of the seeded kernel with every array displaced by 0 / 4 / 8 / 12 bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned v4 __attribute__((ext_vector_type(4)));
typedef v4 v4u __attribute__((aligned(4)));

__global__ __launch_bounds__(256) void k(const v4u *a, v4u *b, v4u *c, size_t nvec)
{
	const size_t stride = (size_t)gridDim.x * 256;
	for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < nvec; g += stride) {
		v4 t = a[g];
		b[g] = t + 1u;
		c[g] = t ^ 0x55u;
	}
}

int main()
{
	const size_t n = (size_t)1 << 28;		// words per array
	unsigned *a, *b, *c;
	hipMalloc((void **)&a, n * 4 + 64); hipMalloc((void **)&b, n * 4 + 64);
	hipMalloc((void **)&c, n * 4 + 64);
	std::vector<unsigned> h(1 << 16);
	for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)(i * 2654435761u);
	for (int off = 0; off < 4; off++) {
		hipMemset(b, 0, n * 4 + 64); hipMemset(c, 0, n * 4 + 64);
		hipMemcpy(a + off, h.data(), h.size() * 4, hipMemcpyHostToDevice);
		const size_t nvec = (n - 4) / 4;
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		k<<<2048, 256>>>((const v4u *)(a + off), (v4u *)(b + off), (v4u *)(c + off), nvec);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		for (int r = 0; r < 10; r++)
			k<<<2048, 256>>>((const v4u *)(a + off), (v4u *)(b + off), (v4u *)(c + off), nvec);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
		std::vector<unsigned> hb(h.size()), hc(h.size());
		hipMemcpy(hb.data(), b + off, h.size() * 4, hipMemcpyDeviceToHost);
		hipMemcpy(hc.data(), c + off, h.size() * 4, hipMemcpyDeviceToHost);
		bool ok = hipGetLastError() == hipSuccess;
		for (size_t i = 0; i < h.size() && ok; i++)
			ok = hb[i] == h[i] + 1u && hc[i] == (h[i] ^ 0x55u);
		printf("offset %2d bytes: %s  %.3f ms  %.1f GB/s\n", off * 4, ok ? "ok" : "WRONG",
			ms, nvec * 16.0 * 3 / ms / 1e6);
	}
	return 0;
}
