// ref_utility_wrap.cpp — extern "C" shims over the REFERENCE's own src/Utility.cpp (validators, SAH cost, OBJ loader).
// TEST INFRASTRUCTURE ONLY.  Compiled together with /root/reference/src/Utility.cpp (unmodified, where it lies) into
// oracle/_ref/libref_utility.so by oracle/Makefile; the reference sources are never copied into this repo.
#include <src/Utility.h>
#include <cstring>
#include <string>
#include <vector>

using namespace BvhConstruction;

extern "C" {

float ref_calculateLbvhCost(const void* nodes, u32 root, u32 nLeaf, u32 nInternal) {
	return Utility::calculateLbvhCost((const Bvh2Node*)nodes, root, nLeaf, nInternal);       // src/Utility.cpp:317-349
}
int ref_checkLbvhRootAabb(const void* nodes, u32 root, u32 nLeaf, u32 nInternal) {
	return Utility::checkLbvhRootAabb((const Bvh2Node*)nodes, root, nLeaf, nInternal) ? 1 : 0;   // :15-27
}
int ref_checkLBvhCorrectness(const void* nodes, u32 root, u32 nLeaf, u32 nInternal) {
	return Utility::checkLBvhCorrectness((const Bvh2Node*)nodes, root, nLeaf, nInternal) ? 1 : 0;   // :31-60 (32-entry stack!)
}
int ref_checkPlocBvh2Correctness(const void* nodes, const void* leaves, u32 root, u32 nLeaf, u32 nInternal) {
	return Utility::checkPlocBvh2Correctness((const Bvh2Node*)nodes, (const PrimRef*)leaves, root, nLeaf, nInternal) ? 1 : 0;   // :62-91
}
int ref_checkLBvh4Correctness(const void* bvh4, const void* primNodes, u32 root, u32 nInternal) {
	return Utility::checkLBvh4Correctness((const Bvh4Node*)bvh4, (const PrimNode*)primNodes, root, nInternal) ? 1 : 0;   // :93-130
}
float ref_calculatebvh4Cost(const void* bvh4, const void* primNodes, void* primAabbs, u32 root, u32 total, u32 nInternal) {
	return Utility::calculatebvh4Cost((const Bvh4Node*)bvh4, (const PrimNode*)primNodes, (Aabb*)primAabbs, root, total, nInternal);   // :351-396
}
float ref_calculateBinnedSahBvhCost(const void* nodes, u32 root, u32 total) {
	return Utility::calculateBinnedSahBvhCost((const SahBvhNode*)nodes, root, total);         // :398-422
}
// MeshLoader::loadScene (src/Utility.cpp:614-760).  Two-call protocol: tris_out == NULL returns the count.
u32 ref_loadScene(const char* obj, const char* mtlDir, void* tris_out, u32 capacity) {
	std::vector<Triangle> t;
	MeshLoader::loadScene(obj, mtlDir, t);
	if (tris_out) std::memcpy(tris_out, t.data(), sizeof(Triangle) * (t.size() < capacity ? t.size() : capacity));
	return (u32)t.size();
}
// Utility::doEarlySplitClipping with the default saMax (src/Utility.cpp:456-538) -> PrimRef[n]
u32 ref_primRefs(const void* tris, u32 n, void* refs_out) {
	std::vector<Triangle> t((const Triangle*)tris, (const Triangle*)tris + n);
	std::vector<PrimRef> r;
	Utility::doEarlySplitClipping(t, r);
	if (refs_out) std::memcpy(refs_out, r.data(), sizeof(PrimRef) * r.size());
	return (u32)r.size();
}
u32 ref_sizeof(int what) {
	switch (what) { case 0: return sizeof(Triangle); case 1: return sizeof(Bvh2Node); case 2: return sizeof(PrimRef); case 3: return sizeof(Aabb);
	                case 4: return sizeof(Bvh4Node); case 5: return sizeof(SahBvhNode); case 6: return sizeof(PrimNode); case 7: return sizeof(Ray); }
	return 0;
}
}
