// lds_swizzle_search.c -- is there a per-column (xor, offset) LDS slot map for the packed column kernel (4 columns side by side, lane = 4 b + c,
// 16-byte elements) that makes BOTH the 16-lane groups of ds_read_b128 and the 8-lane groups of ds_write_b128 conflict-free?  Exhaustive over
// slot = ((K + b) ^ x_c) + o_c, x_c, o_c in 0..15: no.  Today (x = 0, o = 4 c: lds_col_stride) = reads conflict-free, writes two-way, the best
// of the family.  gcc -O2 tools/exp/lds_swizzle_search.c && ./a.out   (MI355X_MICROARCH.md, LDS table: lane groups and bank formulas)
#include <stdio.h>
#include <string.h>
static const int G1[16]={0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27}, G2[16]={4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31};
int check(const int *x, const int *o, int *wr, int *ww){
    int worst_r=1, worst_w=1;
    for(int K=0;K<16;++K){
        int slot[64];
        for(int l=0;l<64;++l){int c=l%4,b=l/4; slot[l]=(((K+b)^x[c])+o[c]);}
        for(int g=0;g<4;++g){ int cnt[16]; memset(cnt,0,sizeof cnt);
            for(int i=0;i<16;++i){int l=(g&1?G2[i]:G1[i])+(g>>1)*32; int m=++cnt[slot[l]&15]; if(m>worst_r)worst_r=m;} }
        for(int g=0;g<8;++g){ int cnt[8]; memset(cnt,0,sizeof cnt);
            for(int i=0;i<8;++i){int m=++cnt[slot[8*g+i]&7]; if(m>worst_w)worst_w=m;} }
        if(worst_r>1 && worst_w>1) break;
    }
    *wr=worst_r; *ww=worst_w; return worst_r==1&&worst_w==1;
}
int main(){
    int x[4]={0,0,0,0},o[4]={0,4,8,12},r,w; check(x,o,&r,&w); printf("current: read %d-way write %d-way\n",r,w);
    long found=0; int best_r=9,best_w=9;
    for(int a=0;a<4096;++a) for(int b=0;b<4096;++b){
        x[1]=a&15;x[2]=(a>>4)&15;x[3]=(a>>8)&15; o[1]=b&15;o[2]=(b>>4)&15;o[3]=(b>>8)&15; x[0]=0;o[0]=0;
        if(check(x,o,&r,&w)){ if(found<8) printf("found x=(0,%d,%d,%d) o=(0,%d,%d,%d)\n",x[1],x[2],x[3],o[1],o[2],o[3]); ++found; }
        if(r==1 && w<best_w){best_w=w; printf("reads free, writes %d-way: x=(0,%d,%d,%d) o=(0,%d,%d,%d)\n",w,x[1],x[2],x[3],o[1],o[2],o[3]);}
    }
    printf("conflict-free maps: %ld\n",found); return 0;
}
