#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f b)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
__device__ __forceinline__ v2f pk_cmulc(v2f a, v2f b)   // conj(a) * b
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
__global__ void k(const float2 *a, const float2 *b, float2 *o)
{
    int i = threadIdx.x;
    v2f x = {a[i].x, a[i].y}, y = {b[i].x, b[i].y};
    v2f r = pk_cmul(x, y), c = pk_cmulc(x, y);
    v2f s;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(s) : "v"(r), "v"(c));
    o[3*i] = make_float2(r.x, r.y); o[3*i+1] = make_float2(c.x, c.y); o[3*i+2] = make_float2(s.x, s.y);
}
int main(){ 
  float2 ha[64], hb[64], ho[192]; for(int i=0;i<64;i++){ha[i]=make_float2(1.5f+i,-2.25f+0.5f*i); hb[i]=make_float2(0.75f-i,3.0f+0.25f*i);} 
  float2 *a,*b,*o; hipMalloc(&a,sizeof(ha)); hipMalloc(&b,sizeof(hb)); hipMalloc(&o,sizeof(ho));
  hipMemcpy(a,ha,sizeof(ha),hipMemcpyHostToDevice); hipMemcpy(b,hb,sizeof(hb),hipMemcpyHostToDevice);
  k<<<1,64>>>(a,b,o); hipMemcpy(ho,o,sizeof(ho),hipMemcpyDeviceToHost);
  int bad=0; for(int i=0;i<64;i++){ float ax=ha[i].x, ay=ha[i].y, bx=hb[i].x, by=hb[i].y;
    float rx=ax*bx-ay*by, ry=ax*by+ay*bx, cx=ax*bx+ay*by, cy=ax*by-ay*bx; float sx=rx+cy, sy=ry-cx;
    if (fabsf(ho[3*i].x-rx)>1e-3f*fabsf(rx)+1e-3f||fabsf(ho[3*i].y-ry)>1e-3f*fabsf(ry)+1e-3f||fabsf(ho[3*i+1].x-cx)>1e-3f*fabsf(cx)+1e-3f||fabsf(ho[3*i+1].y-cy)>1e-3f*fabsf(cy)+1e-3f||fabsf(ho[3*i+2].x-sx)>1e-2f||fabsf(ho[3*i+2].y-sy)>1e-2f) bad++; }
  printf("bad %d  sample r=(%g,%g) c=(%g,%g) s=(%g,%g)\n", bad, ho[3].x, ho[3].y, ho[4].x, ho[4].y, ho[5].x, ho[5].y); return bad; }
