// threading/Threading.h:32-51 of the reference for the chisel_hip facade: chisel::parallel_for, the host-side helper the reference's
// frame driver fans its chunks out with (Chisel.h:150-195, 16 threads).  The facade's own frame driver has no use for it -- a launch set
// on the GPU takes that place -- but third-party callers of the header keep compiling: same signature, same partition (groups of
// max(|threshold|, n / |nthreads|) consecutive elements, one thread per group, the last group on the calling thread).
#ifndef CHISEL_HIP_FACADE_THREADING_H_
#define CHISEL_HIP_FACADE_THREADING_H_
#include <algorithm>
#include <cassert>
#include <cstddef>
#include <cstdlib>
#include <thread>
#include <vector>

namespace chisel {
template <typename Iterator, class Function>
void parallel_for(const Iterator &first, const Iterator &last, Function &&f, const int nthreads = 16, const int threshold = 1000) {
    const std::ptrdiff_t n = last - first;
    const std::ptrdiff_t group = std::max<std::ptrdiff_t>(std::max<std::ptrdiff_t>(1, std::abs(threshold)), n / std::abs(nthreads));
    std::vector<std::thread> workers;
    workers.reserve((size_t)std::abs(nthreads));
    Iterator it = first;
    while (it < last - group) {
        const Iterator stop = std::min(it + group, last);
        workers.emplace_back([it, stop, &f]() { std::for_each(it, stop, f); });
        it = stop;
    }
    std::for_each(it, last, f);  // the remainder, while the others run
    for (std::thread &w : workers) w.join();
}
}  // namespace chisel
#endif
