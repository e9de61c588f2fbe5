// build_example.cpp — the reference's driver (src/main.cpp:52-65) re-expressed against include/bvh/builders.hpp.
// Usage: build_example [n_triangles] [builder: two|single|ploc|hploc]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bvh/builders.hpp"

using namespace BvhConstruction;

static std::vector<Triangle> makeMesh(size_t n) {   // small deterministic soup (LCG), enough to exercise the API
    std::vector<Triangle> t(n);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1103515245u + 12345u; return float(s >> 8) / 16777216.0f; };
    for (auto& tri : t) {
        const float3 c{rnd(), rnd(), rnd()};
        float3* v[3] = {&tri.v1, &tri.v2, &tri.v3};
        for (auto* p : v) *p = float3{c.x + (rnd() - .5f) * .02f, c.y + (rnd() - .5f) * .02f, c.z + (rnd() - .5f) * .02f};
    }
    return t;
}

template <typename B> static int run(Context& ctx, std::vector<Triangle>& tris, const char* name) {
    B bvh;
    bvh.build(ctx, tris);
    std::printf("== %s: %zu triangles, root %u, %u internal nodes, %u wide nodes, BVH2 SAH %g\n", name, tris.size(), bvh.m_rootNodeIdx, bvh.m_nInternalNodes, bvh.m_nWideNodes, bvh.m_costBvh2);
    std::fflush(stdout);
    bvh.traverseBvh(ctx);
    const auto nodes = bvh.d_bvhNodes.getData();
    const Aabb& r = nodes[bvh.m_rootNodeIdx].m_aabb;
    std::printf("root aabb [%g %g %g] - [%g %g %g]\n", r.m_min.x, r.m_min.y, r.m_min.z, r.m_max.x, r.m_max.y, r.m_max.z);
    return 0;
}

int main(int argc, char** argv) {
    try {
        const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 100000;
        const char* which = argc > 2 ? argv[2] : "hploc";
        Context context;
        auto tris = makeMesh(n);
        if (!std::strcmp(which, "two")) return run<TwoPassLbvh>(context, tris, "TwoPassLbvh");
        if (!std::strcmp(which, "single")) return run<SinglePassLbvh>(context, tris, "SinglePassLbvh");
        if (!std::strcmp(which, "ploc")) return run<PLOCNew>(context, tris, "PLOCNew");
        return run<HPLOC>(context, tris, "HPLOC");
    } catch (std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
