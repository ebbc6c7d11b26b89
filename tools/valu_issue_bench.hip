// tools/valu_issue_bench.hip -- diagnostic (not part of the library): how fast does one gfx950 SIMD issue wave64 VALU
// instructions, as a function of the waves resident on it and of the instruction mix?  Settles the "2 or 4 cycles per
// wave64 op" question behind DESIGN.md section 3.1 (MI355X_MICROARCH.md:54,473,489).
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue_bench tools/valu_issue_bench.hip && ./valu_issue_bench
//
// Every workgroup has 256 threads = one wave per SIMD of its CU; k workgroups per CU give k waves per SIMD (the
// dynamic LDS request pins the residency: 160 KiB / k per workgroup).  Each wave runs ITER iterations of a block of
// independent (or dependent) instructions; cycles per wave-instruction per SIMD = elapsed shader cycles * 1 /
// (instructions per wave * k).  The shader clock is read in-kernel (s_memtime against s_memrealtime at 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long t0, t1, r0, r1; };

// MODE 0: 16 independent v_fma_f32 per iteration; 1: one dependent chain of v_fma_f32; 2: 8 independent fma + 8 v_exp_f32
// (transcendental rate); 3: 16 independent fma + 4 ds_read_b32 round trips (an LDS wait every 4 fma); 4: v_pk_fma_f32
template <int MODE>
__global__ __launch_bounds__(256) void issue_kernel(float* out, Stamp* st, int iters, float a, float b) {
	extern __shared__ float lds[];
	float r[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) r[i] = (float)(threadIdx.x + i) * 1e-3f;
	lds[threadIdx.x] = a;
	__syncthreads();
	const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < iters; ++it) {
		if (MODE == 0) {
#pragma unroll
			for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
		} else if (MODE == 1) {
#pragma unroll
			for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(a), "v"(b));
		} else if (MODE == 2) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
				asm volatile("v_exp_f32 %0, %0" : "+v"(r[8 + i]));
			}
		} else if (MODE == 3) {
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				float v;
				asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((threadIdx.x & 255) * 4) : "memory");
#pragma unroll
				for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[4 * g + i]) : "v"(v), "v"(b));
			}
		} else if (MODE >= 5) {
			// one instruction class per mode, 16 independent instructions per iteration
			int* ri = (int*)r;
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				if (MODE == 5) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(a));
				if (MODE == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(ri[i]) : "v"(it));
				if (MODE == 7) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
				if (MODE == 8) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[i]));
				if (MODE == 9) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
				if (MODE == 10) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(ri[i]) : "v"(it));
				if (MODE == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ri[i]) : "v"(it));
				if (MODE == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
				if (MODE == 13) { int sg; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(ri[i])); asm volatile("v_add_u32 %0, %0, %1" : "+v"(ri[(i + 8) & 15]) : "s"(sg)); }
				if (MODE == 14) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(((double*)r)[i & 7]) : "v"((double)a), "v"((double)b));
				if (MODE == 15) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
				if (MODE == 16) asm volatile("s_add_u32 %0, %0, 1\n\tv_add_f32 %1, %1, %2" : "+s"(it) , "+v"(r[i]) : "v"(a)); // never executed: placeholder
			}
		} else {
			typedef float f2 __attribute__((ext_vector_type(2)));
			f2* p = (f2*)r;
			const f2 aa = {a, a}, bb = {b, b};
#pragma unroll
			for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(aa), "v"(bb));
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	float s = 0.f;
#pragma unroll
	for (int i = 0; i < 16; ++i) s += r[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
	if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t0, t1, r0, r1};
}

template <int MODE>
static void run(const char* name, int instPerIter, int nCU, float* dOut, Stamp* dSt, int iters) {
	for (int k : {1, 2, 4, 8}) {
		const int blocks = nCU * k;
		const size_t ldsBytes = (size_t)(160 * 1024 / k) & ~(size_t)1023; // k workgroups fit a CU, k + 1 do not
		CHK(hipFuncSetAttribute((const void*)issue_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
		hipEvent_t e0, e1;
		CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
		hipLaunchKernelGGL(issue_kernel<MODE>, dim3(blocks), dim3(256), ldsBytes, 0, dOut, dSt, iters / 8, 1.0001f, 1e-7f); // warm
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL(issue_kernel<MODE>, dim3(blocks), dim3(256), ldsBytes, 0, dOut, dSt, iters, 1.0001f, 1e-7f);
		CHK(hipEventRecord(e1));
		CHK(hipDeviceSynchronize());
		float ms = 0.f;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		std::vector<Stamp> st(blocks);
		CHK(hipMemcpy(st.data(), dSt, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost));
		std::vector<double> cyc, ghz;
		for (auto& s : st) { cyc.push_back((double)(s.t1 - s.t0)); ghz.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1); }
		std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
		const double medCyc = cyc[cyc.size() / 2], medGHz = ghz[ghz.size() / 2];
		const double perWave = medCyc / ((double)iters * instPerIter);   // cycles a wave needs per instruction
		const double perSimd = perWave / k;                              // SIMD cycles per wave-instruction
		// by the wall clock of the launch: all k waves of a SIMD are resident from start to end only if the dispatcher packed
		// k workgroups onto every CU at once; the wall-clock figure is the conservative one
		const double wallCyc = (double)ms * 1e-3 * medGHz * 1e9 / ((double)iters * instPerIter * k);
		printf("{\"mode\": \"%s\", \"waves_per_simd\": %d, \"kernel_ms\": %.3f, \"clock_GHz\": %.3f, \"cycles_per_inst_per_wave\": %.2f, "
		       "\"simd_cycles_per_wave_inst\": %.2f, \"simd_cycles_per_wave_inst_wall\": %.2f}\n", name, k, ms, medGHz, perWave, perSimd, wallCyc);
		CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
	}
}

// copy bandwidth: float4 grid-stride copy of 1 GiB
__global__ void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main() {
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int nCU = prop.multiProcessorCount;
	printf("{\"device\": \"%s\", \"arch\": \"%s\", \"CUs\": %d, \"clockRate_kHz\": %d}\n", prop.name, prop.gcnArchName, nCU, prop.clockRate);
	float* dOut; Stamp* dSt;
	CHK(hipMalloc(&dOut, sizeof(float) * 256 * nCU * 8));
	CHK(hipMalloc(&dSt, sizeof(Stamp) * nCU * 8));
	const int iters = 200000;
	run<0>("16 independent v_fma_f32", 16, nCU, dOut, dSt, iters);
	run<1>("dependent v_fma_f32 chain", 16, nCU, dOut, dSt, iters);
	run<2>("8 v_fma_f32 + 8 v_exp_f32", 16, nCU, dOut, dSt, iters);
	run<3>("16 v_fma_f32 + 4 ds_read_b32 waits", 20, nCU, dOut, dSt, iters / 4);
	run<4>("8 independent v_pk_fma_f32", 8, nCU, dOut, dSt, iters);
	run<5>("v_add_f32 dpp quad_perm", 16, nCU, dOut, dSt, iters);
	run<6>("v_and_b32", 16, nCU, dOut, dSt, iters);
	run<7>("v_med3_f32", 16, nCU, dOut, dSt, iters);
	run<8>("v_cvt_i32_f32", 16, nCU, dOut, dSt, iters);
	run<9>("v_cmp_lt_f32 + v_cndmask_b32 (pairs)", 32, nCU, dOut, dSt, iters);
	run<10>("v_mul_u32_u24", 16, nCU, dOut, dSt, iters);
	run<11>("v_mul_lo_u32", 16, nCU, dOut, dSt, iters);
	run<12>("v_mul_f32", 16, nCU, dOut, dSt, iters);
	run<13>("v_readlane_b32 + v_add_u32 sgpr (pairs)", 32, nCU, dOut, dSt, iters / 2);
	run<14>("v_fma_f64", 16, nCU, dOut, dSt, iters / 2);
	run<15>("v_rcp_f32", 16, nCU, dOut, dSt, iters / 2);
	// copy bandwidth
	const size_t bytes = (size_t)1 << 30;
	float4 *a, *b;
	CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes));
	CHK(hipMemset(a, 1, bytes));
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	for (int rep = 0; rep < 3; ++rep) {
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL(copy_kernel, dim3(nCU * 16), dim3(256), 0, 0, a, b, bytes / 16);
		CHK(hipEventRecord(e1));
		CHK(hipDeviceSynchronize());
		float ms = 0.f;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		printf("{\"copy_GiB\": 1, \"ms\": %.3f, \"read_plus_write_GB_per_s\": %.1f}\n", ms, 2.0 * bytes / ms * 1e-6);
	}
	return 0;
}
