#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdint>
int main(){ for(int T:{1,2,4,8}){ auto t0=std::chrono::steady_clock::now(); std::vector<std::thread> th; volatile uint64_t sink[64]={0};
 for(int t=0;t<T;t++) th.emplace_back([&,t]{ uint64_t x=t+1; for(long i=0;i<200000000L/T;i++){ x=x*6364136223846793005ULL+1442695040888963407ULL;} sink[t*8]=x;});
 for(auto&x:th)x.join(); printf("T=%d %.1f ms\n",T,std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count()); } }
