#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
static int ref_find(uint64_t bits, int n, int v){ int i; for(i=0;i<n;i++) if(((bits>>(4*i))&15)==(uint64_t)v) break; return i; }
static int fast_find(uint64_t bits, int n, int v){
  uint64_t x = bits ^ (0x1111111111111111ull * (uint64_t)v);
  uint64_t t = (x - 0x1111111111111111ull) & ~x & 0x8888888888888888ull;
  if (n < 16) t &= (((uint64_t)1 << (4*n)) - 1);
  return t ? (__builtin_ctzll(t) >> 2) : n;
}
int main(){ srand(1); long bad=0; for(long k=0;k<200000000;k++){ uint64_t b=((uint64_t)rand()<<40)^((uint64_t)rand()<<20)^rand(); int n=rand()%17; int v=rand()%16; if(ref_find(b,n,v)!=fast_find(b,n,v)){bad++; if(bad<5) printf("bad %llx %d %d: %d %d\n",(unsigned long long)b,n,v,ref_find(b,n,v),fast_find(b,n,v));}} printf("bad=%ld\n",bad); return 0; }
