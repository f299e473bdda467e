#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
static uint64_t s=88172645463325252ull; static uint64_t rnd(){ s^=s<<13; s^=s>>7; s^=s<<17; return s; }
int main(){ long bad=0, N=1000000000; const double b=3.0, y=1.0/3.0; for(long k=0;k<N;k++){ uint64_t ab=((uint64_t)(1023-40+(rnd()%45))<<52)|(rnd()>>12)|((rnd()&1)<<63); double a; memcpy(&a,&ab,8);
   double q0=a*y; double r=fma(-q0,b,a); double q=fma(r,y,q0); double t=a/b; if(q!=t){bad++; if(bad<10) printf("bad a=%a q=%a t=%a\n",a,q,t);} }
 printf("bad=%ld of %ld\n",bad,N); return 0; }
