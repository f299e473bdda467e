#include <stdio.h>
#include <stdint.h>
int main(){
  uint32_t ds[3]={177,176,178};
  for(int k=0;k<3;k++){
    uint32_t d=ds[k];
    uint64_t m=(((uint64_t)1<<32)*(256-d))/d+1;
    printf("d=%u m'=%llu (0x%llx)\n",d,(unsigned long long)m,(unsigned long long)m);
    uint32_t mg=(uint32_t)m; uint64_t bad=0;
    for(uint64_t n=0;n<((uint64_t)1<<32);n++){
      uint32_t x=(uint32_t)n;
      uint32_t t1=(uint32_t)(((uint64_t)mg*x)>>32);
      uint32_t q=(t1+((x-t1)>>1))>>7;
      if(q!=x/d){bad++; if(bad<5)printf("bad %u\n",x);}
    }
    printf("bad=%llu\n",(unsigned long long)bad);
  }
}
