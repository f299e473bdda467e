#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
static uint64_t s=88172645463325252ull; static uint64_t rnd(){ s^=s<<13; s^=s>>7; s^=s<<17; return s; }
int main(){ long bad=0, N=400000000; for(long k=0;k<N;k++){ uint64_t m=rnd(); int e = 1023 - 20 + (int)(rnd()%12); /* theta ~ 1e-6..1e-3 */
   uint64_t bits=((uint64_t)e<<52)|(m>>12); double b; memcpy(&b,&bits,8);
   if ((k & 1023)==0) { bits = ((uint64_t)e<<52)|0xFFFFFFFFFFFFFull; if (k&1024) bits -= (rnd()%4); memcpy(&b,&bits,8);} /* near all-ones significands too */
   double a = (double)(2*(int)(rnd()%70)); if ((k&7)==0) { uint64_t ab=((uint64_t)(1023-3+(rnd()%8))<<52)|(rnd()>>12); memcpy(&a,&ab,8);} /* also generic numerators */
   double y = 1.0/b; double q0=a*y; double r=fma(-q0,b,a); double q=fma(r,y,q0); double t=a/b;
   if(q!=t){bad++; if(bad<10) printf("bad a=%a b=%a q=%a t=%a\n",a,b,q,t);} }
 printf("bad=%ld of %ld\n",bad,N); return 0; }
