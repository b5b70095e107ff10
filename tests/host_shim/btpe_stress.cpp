// Stress of the guarded fast BTPE (csrc/npy_rng.h: binomial_btpe_fast) against the exact restatement of numpy's BTPE on random
// (n, p, generator state): whenever the fast path decides, the draw AND the generator state afterwards must be identical.
// Built twice by tests/test_npy_rng_host.py: plain, and with -DNPY_HOST_PERTURB (every cheap primitive carries an error 2-4x what the
// hardware instruction may have).  usage: btpe_stress <draws>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <random>
static long fb[2]={0,0};
#define NPY_NOTE_FALLBACK(w) (fb[w]++)
#include "npy_rng.h"
int main(int argc,char**argv){
  long N = argc>1? atol(argv[1]) : 50000000;
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> u01(0,1);
  long mism=0, done=0, fast_ok=0;
  for(long it=0; it<N; it++){
    int64_t n; double p;
    int mode = it%5;
    if(mode==0){ n=(int64_t)std::pow(10.0,1.8+4.2*u01(rng)); p=0.5*u01(rng); }
    else if(mode==1){ n=(int64_t)std::pow(10.0,3+6*u01(rng)); double np_=30.0+std::pow(10.0,4*u01(rng)); p=np_/n; }
    else if(mode==2){ n=40000+(int64_t)(20000*u01(rng)); p=std::pow(10.0,-3.2+2.9*u01(rng)); }
    else if(mode==3){ n=(int64_t)(61+1000*u01(rng)); p=0.5-0.5*std::pow(u01(rng),3); }
    else { n=(int64_t)std::pow(10.0,1.8+5.2*u01(rng)); p=(30.0+120.0*u01(rng))/(double)n; }   // n p in [30, 150]: small m, the log form of the explicit test near its limits
    if(p>0.5) p=0.5; if(n>= (1LL<<31)-2) continue; if(p*(double)n<=30.0) continue;
    npyrng::Pcg64 g0{rng(),rng(),rng(),rng()|1};
    npyrng::Pcg64 a=g0,b=g0;
    #ifdef STRESS_LAZY   // the variant the one-chain-per-wave kernel instantiates (rare-branch set-up kept inside the branches)
    int32_t yf = npyrng::binomial_btpe_fast<int32_t, true>(a,(int32_t)n,p,0);
#else
    int32_t yf = npyrng::binomial_btpe_fast<int32_t>(a,(int32_t)n,p,0);
#endif
    int32_t ye = npyrng::binomial_btpe<int32_t>(b,(int32_t)n,p);
    done++;
    if(yf>=0){ fast_ok++; if(yf!=ye || a.s_hi!=b.s_hi || a.s_lo!=b.s_lo){ mism++; if(mism<10) printf("MISMATCH n=%ld p=%.17g fast=%d exact=%d\n",(long)n,p,yf,ye);} }
  }
  printf("draws %ld fast-decided %ld (%.4f) mismatches %ld\n",done,fast_ok,(double)fast_ok/done,mism);
  return mism!=0;
}
