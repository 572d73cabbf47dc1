#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <time.h>
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+1e-9*t.tv_nsec;}
int main(){ size_t n=400u<<20; 
 char*a=mmap(0,n+(2<<20),PROT_READ|PROT_WRITE,MAP_PRIVATE|MAP_ANONYMOUS,-1,0); double t=now(); for(size_t i=0;i<n;i+=4096)a[i]=1; printf("plain: %.1f ms\n",(now()-t)*1e3);
 char*b=mmap(0,n+(2<<20),PROT_READ|PROT_WRITE,MAP_PRIVATE|MAP_ANONYMOUS,-1,0); b=(char*)(((size_t)b+(2<<20)-1)&~(size_t)((2<<20)-1)); madvise(b,n,MADV_HUGEPAGE); t=now(); for(size_t i=0;i<n;i+=4096)b[i]=1; printf("hugepage: %.1f ms\n",(now()-t)*1e3);
 char*c=mmap(0,n,PROT_READ|PROT_WRITE,MAP_PRIVATE|MAP_ANONYMOUS|MAP_POPULATE,-1,0); t=now(); for(size_t i=0;i<n;i+=4096)c[i]=1; printf("populate (touch after): %.1f ms\n",(now()-t)*1e3);
 t=now(); char*d=mmap(0,n,PROT_READ|PROT_WRITE,MAP_PRIVATE|MAP_ANONYMOUS|MAP_POPULATE,-1,0); printf("populate mmap itself: %.1f ms\n",(now()-t)*1e3); (void)d;
 return 0;}
