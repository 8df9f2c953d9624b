// Ad-hoc (round 5, CPU): how long until a Whittaker substitution chain started from a zero state mid-row gives the row's own
// values bit for bit (two in a row: from there on for good) -- six kinds of rows, both sweeps, 200 starts each.
//   gcc -O2 -ffp-contract=off -o coalesce scripts/ubench/coalesce.c oracle/baseline_oracle.c -Ioracle -lm && ./coalesce [block]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdint.h>
void oracle_whittaker_factor_f64(size_t n, int parity, double lambda, double *d, double *l1, double *l2);
static uint64_t st=88172645463325252ULL; static double urand(){ st^=st<<13; st^=st>>7; st^=st<<17; return ((st>>11)+0.5)/9007199254740992.0; }
static int cmp(const void*a,const void*b){ long x=*(const long*)a,y=*(const long*)b; return (x>y)-(x<y);}
int main(int argc,char**argv){
  size_t n=2000000; double w=101*0.15915494; double lambda=7.0*w*w*w*w;
  if(argc>1){ double blk=atof(argv[1]); w=blk*0.15915494; lambda=7.0*w*w*w*w; }
  double *d=malloc(8*n),*l1=malloc(8*n),*l2=malloc(8*n),*y=malloc(8*n),*f=malloc(8*n),*g=malloc(8*n),*z=malloc(8*n),*x=malloc(8*n);
  const char*kinds[]={"poisson3 log2 centred","sparse peaks (95% zero counts)","gaussian","constant -1.5","poisson30 log2 centred","multiplied by normal (null draw)"};
  for(int kind=0;kind<6;kind++){
   for(int parity=0;parity<2;parity++){
    oracle_whittaker_factor_f64(n,parity,lambda,d,l1,l2);
    for(size_t i=0;i<n;i++){ double v;
      if(kind==0){ double u=urand(); double c=floor(-log(u)*3.0); v=log2(c+1.0)-1.7; }
      else if(kind==1){ double u=urand(); double c=(u<0.95)?0.0:floor(-log(urand())*8.0); v=log2(c+1.0)-0.0; }
      else if(kind==2){ double u1=urand(),u2=urand(); v=sqrt(-2*log(u1))*cos(6.283185307179586*u2); }
      else if(kind==3){ v=-1.5; }
      else if(kind==4){ double u=urand(); double c=floor(30.0+sqrt(30.0)*sqrt(-2*log(u))*cos(6.283185307179586*urand())); if(c<0)c=0; v=log2(c+1.0)-4.9; }
      else { double u1=urand(),u2=urand(); double c=floor(-log(urand())*3.0); v=(log2(c+1.0)-1.7)*sqrt(-2*log(u1))*cos(6.283185307179586*u2); }
      y[i]=v; }
    for(size_t i=0;i<n;i++){ int mine=((i&1)==(size_t)parity); g[i]=(i<2||i+2>=n)?(mine?y[i]:0.0):(mine?1.0:0.0)*y[i]; }
    memcpy(f,g,8*n);
    f[1]=f[1]-l1[0]*f[0];
    for(size_t i=2;i<n;i++){ double t1=l1[i-1]*f[i-1], t2=l2[i-2]*f[i-2]; f[i]=f[i]-t1-t2; }
    for(size_t i=0;i<n;i++) z[i]=f[i]/d[i];
    x[n-1]=z[n-1]; x[n-2]=z[n-2]-l1[n-2]*x[n-1];
    for(size_t i=n-2;i-->0;){ double t1=l1[i]*x[i+1], t2=l2[i]*x[i+2]; x[i]=z[i]-t1-t2; }
    long times[400]; int nt=0, never=0; const long LIM=400000;
    for(int dir=0;dir<2;dir++){
     nt=0;never=0;
     for(int trial=0;trial<200;trial++){
      size_t s=500000+(size_t)(urand()*1000000);
      double p1=0,p2=0; long first=-1; int run=0;
      if(dir==0){ for(size_t i=s;i<s+LIM;i++){ double t1=l1[i-1]*p1,t2=l2[i-2]*p2; double r=g[i]-t1-t2; p2=p1;p1=r; if(r==f[i]&&(1.0/r==1.0/f[i]||r!=0)){run++; if(run==2){first=i-s;break;}} else run=0; } }
      else { for(size_t i=s;i>s-LIM;i--){ double t1=l1[i]*p1,t2=l2[i]*p2; double r=z[i]-t1-t2; p2=p1;p1=r; if(r==x[i]){run++; if(run==2){first=s-i;break;}} else run=0; } }
      if(first<0) never++; else times[nt++]=first;
     }
     qsort(times,nt,sizeof(long),cmp);
     double mean=0; for(int i=0;i<nt;i++) mean+=times[i]; mean/= (nt?nt:1);
     printf("%-34s parity %d %s: coalesced %3d/200  mean %8.0f  median %7ld  p90 %7ld  p99 %7ld  max %7ld  never(<%ld) %d\n",kinds[kind],parity,dir?"backward":"forward ",nt,mean,nt?times[nt/2]:-1,nt?times[(int)(nt*0.9)]:-1,nt?times[(int)(nt*0.99)]:-1,nt?times[nt-1]:-1,LIM,never);
    }
   }
  }
  return 0;
}
