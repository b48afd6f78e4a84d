#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
// A: dword per lane, 1 per iteration, grid-stride
__global__ void rd1(const float* __restrict__ p, int64_t n, float* out){
  float s=0; for(int64_t i=(int64_t)blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=(int64_t)gridDim.x*blockDim.x) s+=p[i];
  if(s==123.456f) out[0]=s;
}
// B: 8 dwords per lane in flight, block handles contiguous 256*8 chunk
__global__ void rd8(const float* __restrict__ p, int64_t n, float* out){
  int64_t base=((int64_t)blockIdx.x*256*8); float v[8];
#pragma unroll
  for(int k=0;k<8;k++){ int64_t i=base+k*256+threadIdx.x; v[k]= i<n? p[i]:0.f; }
  float s=0;
#pragma unroll
  for(int k=0;k<8;k++) s+=v[k];
  if(s==123.456f) out[0]=s;
}
// C: float4 per lane, 2 in flight
__global__ void rd4x(const float4* __restrict__ p, int64_t n4, float* out){
  int64_t base=((int64_t)blockIdx.x*256*2); float4 v[2];
#pragma unroll
  for(int k=0;k<2;k++){ int64_t i=base+k*256+threadIdx.x; v[k]= i<n4? p[i]:make_float4(0,0,0,0); }
  float s=v[0].x+v[0].y+v[0].z+v[0].w+v[1].x+v[1].y+v[1].z+v[1].w;
  if(s==123.456f) out[0]=s;
}
// D: dword + ushort streams, 8 in flight each (the mc_bits pattern)
__global__ void rd8_us(const float* __restrict__ p, const unsigned short* __restrict__ q, int64_t n, float* out){
  int64_t base=((int64_t)blockIdx.x*256*8); float v[8]; int c[8];
#pragma unroll
  for(int k=0;k<8;k++){ int64_t i=base+k*256+threadIdx.x; v[k]= i<n? p[i]:0.f; c[k]= i<n? q[i]:0; }
  float s=0;
#pragma unroll
  for(int k=0;k<8;k++) s+=v[k]+c[k];
  if(s==123.456f) out[0]=s;
}
// E: float4 + ushort4 per lane
__global__ void rd4_us4(const float4* __restrict__ p, const ushort4* __restrict__ q, int64_t n4, float* out){
  int64_t base=((int64_t)blockIdx.x*256*2); float s=0;
  float4 v[2]; ushort4 c[2];
#pragma unroll
  for(int k=0;k<2;k++){ int64_t i=base+k*256+threadIdx.x; if(i<n4){v[k]=p[i]; c[k]=q[i];} else {v[k]=make_float4(0,0,0,0); c[k]=make_ushort4(0,0,0,0);} }
#pragma unroll
  for(int k=0;k<2;k++) s+=v[k].x+v[k].y+v[k].z+v[k].w+c[k].x+c[k].y+c[k].z+c[k].w;
  if(s==123.456f) out[0]=s;
}
int main(){
  const int64_t n=(int64_t)1<<30; float* p; unsigned short* q; float* out;
  CK(hipMalloc(&p,n*4)); CK(hipMalloc(&q,n*2)); CK(hipMalloc(&out,4)); CK(hipMemset(p,0,n*4)); CK(hipMemset(q,0,n*2));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); float ms;
  for(int rep=0;rep<2;rep++){
    hipEventRecord(a); hipLaunchKernelGGL(rd1,dim3(256*32),dim3(256),0,0,p,n,out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); printf("rd1 dword gridstride: %.3f ms %.2f TB/s\n",ms,n*4/ms/1e9);
    hipEventRecord(a); hipLaunchKernelGGL(rd8,dim3(n/2048),dim3(256),0,0,p,n,out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); printf("rd8 dword x8 in flight: %.3f ms %.2f TB/s\n",ms,n*4/ms/1e9);
    hipEventRecord(a); hipLaunchKernelGGL(rd4x,dim3(n/4/512),dim3(256),0,0,(const float4*)p,n/4,out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); printf("rd4x float4 x2: %.3f ms %.2f TB/s\n",ms,n*4/ms/1e9);
    hipEventRecord(a); hipLaunchKernelGGL(rd8_us,dim3(n/2048),dim3(256),0,0,p,q,n,out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); printf("rd8_us dword+ushort x8: %.3f ms %.2f TB/s\n",ms,n*6/ms/1e9);
    hipEventRecord(a); hipLaunchKernelGGL(rd4_us4,dim3(n/4/512),dim3(256),0,0,(const float4*)p,(const ushort4*)q,n/4,out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); printf("rd4_us4 float4+ushort4 x2: %.3f ms %.2f TB/s\n",ms,n*6/ms/1e9);
  }
  return 0;
}
