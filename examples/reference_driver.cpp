// reference_driver.cpp — the body of the reference's driver (src/main.cpp:24-83) compiled against include/bvh/builders.hpp:
//
//     Context context;  std::vector<Triangle> triangles;  <load mesh>;
//     X bvh;  bvh.build(context, triangles);  bvh.traverseBvh(context);
//
// for X = TwoPassLbvh / SinglePassLbvh / PLOCNew / HPLOC (the reference picks one with a #define, here argv[1]) and the batched builder.
// The mesh comes from a raw little-endian f32 file of n x 9 floats (tests/golden/*.tri) instead of MeshLoader::loadScene — OBJ parsing is
// outside the hot path.  With a third argument everything the builder exposes is dumped to that file so that
// tests/test_cpp_mirror.py can diff it with the ctypes path: this executable is RUN on the GPU box, not only compiled.
//
// Usage: reference_driver <two|single|ploc|hploc|batched> <mesh.tri> [dump.bin]
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>
#include "bvh/builders.hpp"

using namespace BvhConstruction;

static std::vector<Triangle> loadTri(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    const size_t bytes = (size_t)f.tellg(); f.seekg(0);
    std::vector<float> raw(bytes / 4); f.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)(raw.size() * 4));
    std::vector<Triangle> t(raw.size() / 9);
    for (size_t i = 0; i < t.size(); ++i) {
        const float* p = raw.data() + 9 * i;
        t[i].v1 = float3{p[0], p[1], p[2]}; t[i].v2 = float3{p[3], p[4], p[5]}; t[i].v3 = float3{p[6], p[7], p[8]};
    }
    return t;
}

template <typename T> static void put(std::ofstream& o, const std::vector<T>& v) {
    const uint64_t n = v.size(), sz = sizeof(T);
    o.write(reinterpret_cast<const char*>(&n), 8); o.write(reinterpret_cast<const char*>(&sz), 8);
    if (n) o.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(n * sz));
}

template <typename B> static int run(Context& context, std::vector<Triangle>& triangles, const char* dump) {
    B bvh;
    bvh.build(context, triangles);
    bvh.traverseBvh(context);
    if (dump) {   // sections: header words, nodes, leaves, sorted keys, sorted values, unsorted keys, values, prim aabbs, scene, triangles, wide nodes, wide leaves, image
        std::ofstream o(dump, std::ios::binary);
        const std::vector<float> head = {(float)bvh.m_rootNodeIdx, (float)bvh.m_nInternalNodes, bvh.m_cost, bvh.m_costBvh2, (float)bvh.m_nWideNodes,
                                         bvh.m_timer.getTimeRecord(CalculateCentroidExtentsTime), bvh.m_timer.getTimeRecord(CalculateMortonCodesTime),
                                         bvh.m_timer.getTimeRecord(SortingTime), bvh.m_timer.getTimeRecord(BvhBuildTime), bvh.m_timer.getTimeRecord(CollapseBvhTime),
                                         (float)bvh.m_width, (float)bvh.m_height};
        put(o, head);
        put(o, bvh.d_bvhNodes.getData()); put(o, bvh.d_leafNodes.getData());
        put(o, bvh.d_sortedMortonCodeKeys.getData()); put(o, bvh.d_sortedMortonCodeValues.getData());
        put(o, bvh.d_mortonCodeKeys.getData()); put(o, bvh.d_mortonCodeValues.getData());
        put(o, bvh.d_triangleAabb.getData()); put(o, bvh.d_sceneExtents.getData()); put(o, bvh.d_triangleBuff.getData());
        put(o, bvh.d_wideBvhNodes.getData(bvh.m_nWideNodes)); put(o, bvh.d_wideLeafNodes.getData());
        put(o, bvh.m_colorBuffer);
    }
    (void)bvh.d_flags.size();           // (src/TwoPassLbvh.h:27: the member exists; its view is empty here)
    {   // a host that keeps ONE reference stage beside the mirror: the sort call of src/Hploc.cpp:63-81 on the builder's own key / value members.  d_mortonCodeValues.ptr()
        // materialises the iota the build never wrote (IotaView); the result must be the build's own sorted arrays.
        const u32 n = (u32)triangles.size();
        DeviceArray<u32> keysOut, valsOut;
        keysOut.resize(context.handle(), n); valsOut.resize(context.handle(), n);
        if (!bvh.d_mortonCodeValues.ptr()) { std::cerr << "d_mortonCodeValues.ptr() is null\n"; return 4; }
        check(bvh_sort_pairs(context.handle(), bvh.d_mortonCodeKeys.ptr(), bvh.d_mortonCodeValues.ptr(), n, keysOut.ptr(), valsOut.ptr(), 0, 32), "bvh_sort_pairs");
        if (keysOut.getData() != bvh.d_sortedMortonCodeKeys.getData() || valsOut.getData() != bvh.d_sortedMortonCodeValues.getData()) { std::cerr << "kept sort stage differs\n"; return 4; }
        std::cout << "kept sort stage: ok" << std::endl;
    }
    return 0;
}

int main(int argc, char* argv[]) {
    try {
        if (argc < 3) { std::cerr << "usage: reference_driver <two|single|ploc|hploc|batched> <mesh.tri> [dump.bin]\n"; return 2; }
        Context context;
        std::vector<Triangle> triangles = loadTri(argv[2]);
        const char* dump = argc > 3 ? argv[3] : nullptr;
        if (!std::strcmp(argv[1], "batched")) {            // src/main.cpp:33-50 (many copies of one mesh)
            BatchedBvhBuilder bvh;
            std::vector<BatchedBuildInput> batches(4);
            for (auto& b : batches) b.m_primitives = triangles;
            bvh.build(context, batches);
            bvh.traverseBvh(context);
            for (size_t m = 1; m < batches.size(); ++m) if (bvh.m_checksums[m] != bvh.m_checksums[0]) { std::cerr << "batched trees differ\n"; return 3; }
            // the reference's public members (src/BatchedBuilder.h:24-30): every mesh's nodes / leaves stay on the device, one root per mesh
            const auto h_bvhNodes = bvh.d_bvhNodes.getData();
            const auto h_leafNodes = bvh.d_primRefs.getData();
            const auto h_roots = bvh.d_rootNodes.getData();
            if (h_roots.size() != batches.size() || bvh.m_rootNodeIdx != h_roots[0] || bvh.m_nInternalNodes != batches.size() * (triangles.size() - 1)) { std::cerr << "batched members\n"; return 4; }
            if (h_bvhNodes.size() != bvh.d_bvhNodes.size() || h_leafNodes.size() != batches.size() * triangles.size()) { std::cerr << "batched arrays\n"; return 5; }
            for (size_t m = 0; m < batches.size(); ++m) {                      // the root's box is the gathered root box; mesh copies are byte-identical
                const Bvh2Node& r = h_bvhNodes[h_roots[m]];
                if (std::memcmp(&r.m_aabb, &bvh.m_rootAabbs[m], sizeof(Aabb)) != 0) { std::cerr << "root box of mesh " << m << "\n"; return 6; }
                if (m && std::memcmp(&h_bvhNodes[h_roots[m]], &h_bvhNodes[h_roots[0]], sizeof(Bvh2Node)) != 0) { std::cerr << "root node of mesh " << m << "\n"; return 7; }
            }
            std::cout << "batched: " << h_bvhNodes.size() << " nodes, " << h_leafNodes.size() << " leaves, BvhBuildTime " << bvh.m_timer.getTimeRecord(BvhBuildTime)
                      << "ms, mean SAH " << bvh.m_cost << ", lanes per device " << bvh.m_lanesPerDevice << std::endl;
            return 0;
        }
        if (!std::strcmp(argv[1], "single")) return run<SinglePassLbvh>(context, triangles, dump);
        if (!std::strcmp(argv[1], "two")) return run<TwoPassLbvh>(context, triangles, dump);
        if (!std::strcmp(argv[1], "ploc")) return run<PLOCNew>(context, triangles, dump);
        if (!std::strcmp(argv[1], "hploc")) return run<HPLOC>(context, triangles, dump);
        std::cerr << "unknown builder " << argv[1] << "\n";
        return 2;
    } catch (std::exception& e) {
        std::cerr << e.what();
        return -1;
    }
}
