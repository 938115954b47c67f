// CPU replay of the device logaddexp (po_device.h PoLaeFast::f): the rint/cvt/ldexp formulation (f_old) against the
// integer-step one the kernels run (f_new), both against a long-double reference.  gcc -O2 -ffp-contract=off check_lae.c -lm
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include "../poreover_amd/csrc/po_lae_tables.h"
static inline uint64_t b(double x){uint64_t u;memcpy(&u,&x,8);return u;}
static inline double fb(uint64_t u){double x;memcpy(&x,&u,8);return x;}
static double f_old(double d){
    double e=0.0;
    if(d>-40.0){
        const double kf=rint(d*PO_64_LN2); const int k=(int)kf; const int j=k&63,m=k>>6;
        double r=fma(-kf,PO_LN2_64_HI,d); r=fma(-kf,PO_LN2_64_LO,r);
        double p=fma(r,1.0/720,1.0/120); p=fma(r,p,1.0/24); p=fma(r,p,1.0/6); p=fma(r,p,0.5); p=fma(r*r,p,r);
        const double th=PO_EXP_T[j][0],tl=PO_EXP_T[j][1];
        e=ldexp(th+fma(th,p,tl),m);
    }
    const double z=1.0+e; const int i=(int)rint((z-1.0)*64.0);
    const double rc=PO_LOG_T[i][0],lh=PO_LOG_T[i][1],ll=PO_LOG_T[i][2];
    const double w=fma(z,rc,-1.0);
    double q=fma(w,1.0/7,-1.0/6); q=fma(w,q,1.0/5); q=fma(w,q,-1.0/4); q=fma(w,q,1.0/3);
    const double s=w*w; double u=fma(s*w,q,ll); u=fma(-0.5,s,u);
    const double res=lh+(w+u);
    return (d==d)?res:-INFINITY;
}
static double f_new(double d){
    double e=0.0;
    if(d>-40.0){
        const double MAGIC=0x1.8p52;
        const double t=fma(d,PO_64_LN2,MAGIC); const double kf=t-MAGIC; const int k=(int)(uint32_t)b(t);
        const int j=k&63;
        double r=fma(-kf,PO_LN2_64_HI,d); r=fma(-kf,PO_LN2_64_LO,r);
        double p=fma(r,1.0/720,1.0/120); p=fma(r,p,1.0/24); p=fma(r,p,1.0/6); p=fma(r,p,0.5); p=fma(r*r,p,r);
        const double th=PO_EXP_T[j][0],tl=PO_EXP_T[j][1];
        const double x=th+fma(th,p,tl);
        e=fb(b(x)+((uint64_t)(int64_t)(k>>6)<<52));
    }
    const double z=1.0+e; const int i=(int)(((uint32_t)(b(z)>>32)-0x3FF00000u+0x2000u)>>14);
    const double rc=PO_LOG_T[i][0],lh=PO_LOG_T[i][1],ll=PO_LOG_T[i][2];
    const double w=fma(z,rc,-1.0);
    double q=fma(w,1.0/7,-1.0/6); q=fma(w,q,1.0/5); q=fma(w,q,-1.0/4); q=fma(w,q,1.0/3);
    const double s=w*w; double u=fma(s*w,q,ll); u=fma(-0.5,s,u);
    return lh+(w+u);
}
static double lae_old(double x1,double x2){int ge=x1>=x2;double hi=ge?x1:x2;double d=ge?(x2-x1):(x1-x2);return hi+f_old(d);}
static double lae_new(double x1,double x2){double hi=fmax(x1,x2),lo=fmin(x1,x2);return hi+f_new(lo-hi);}
int main(int argc,char**argv){
    srand48(7);
    double mo=0,mn=0,mg=0; long nd=0,N=(argc>1)?atol(argv[1]):20000000;
    for(long it=0;it<N;++it){
        double d;
        int c=it%4;
        if(c==0) d=-40.0*drand48(); else if(c==1) d=-drand48(); else if(c==2) d=-exp(-20*drand48()); else d=-45*drand48()*drand48();
        long double ref=log1pl(expl((long double)d));
        double a=f_old(d),n=f_new(d),g=log1p(exp(d));
        double eo=fabsl(a-ref),en=fabsl(n-ref),eg=fabsl(g-ref);
        if(eo>mo)mo=eo; if(en>mn)mn=en; if(eg>mg)mg=eg;
        if(a!=n)nd++;
    }
    printf("max abs err: old %.3g new %.3g glibc-double %.3g; old!=new in %ld of %ld\n",mo,mn,mg,nd,N);
    // special values
    double sp[]={0.0,-0.0,-1e-300,-40.0,-39.9999,-1000.0,-INFINITY,NAN,-0.6931471805599453};
    for(int i=0;i<9;++i) printf("d=%g old=%.17g new=%.17g\n",sp[i],f_old(sp[i]),f_new(sp[i]));
    printf("lae(-inf,-inf) old %g new %g; lae(-inf,-3) old %g new %g; lae(-2,-2) old %.17g new %.17g\n",lae_old(-INFINITY,-INFINITY),lae_new(-INFINITY,-INFINITY),lae_old(-INFINITY,-3),lae_new(-INFINITY,-3),lae_old(-2,-2),lae_new(-2,-2));
    return 0;
}
