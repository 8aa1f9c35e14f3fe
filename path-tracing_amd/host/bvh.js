'use strict';
/*
 * BVH builder + flattener -- host-side counterpart of /root/reference/classes/bvh.js:6-307 and
 * classes/bvhNode.js:4-30, emitting the same arrays the reference uploads as data textures
 * (bvh.js:271-306, main.js:187-190), de-triplicated and typed.
 *
 * Semantics kept (SURVEY.md 8a a16-a19):
 *  - one seed leaf per triangle, box = min/max of its three corners (bvh.js:19-31), centroid = centre
 *    of that box (bvhNode.js:7-8);
 *  - buildBVH(list, depth, 5): <= 5 triangles -> leaf (union of boxes, concatenated index lists);
 *    else STABLE in-place sort by centroid[depth % 3], split at floor(len/2), left then right
 *    (bvh.js:74-118).  The children receive slices of the sorted list, so ties keep the order the
 *    parent's sort left them in;
 *  - pre-order flatten (bvh.js:149-159); miss link = pre-order index of the first node after the
 *    node's subtree, -1 when there is none (net effect of bvh.js:161-187 + 233-239);
 *  - inorderTrianglesIndicesArray = leaf index lists concatenated in pre-order; leaf.first = its
 *    offset there, leaf.count = its length; internal nodes export first = -2, count = 0
 *    (bvh.js:241-269, 285-293).
 * What is NOT kept: the O(n^2) indexOf passes and the global static nodeId counter -- miss links and
 * leaf offsets are produced directly during the pre-order walk (O(n log n) overall).
 * All arithmetic is f64 as in the reference; rounding to f32 happens once, in toTyped().
 */

const LEAF_SIZE = 5; // bvh.js:40

class BVH {
  /** trianglesArray: [{ triangle: {a,b,c: {x,y,z}} , ...}] exactly like main.js:180 passes. */
  constructor(trianglesArray) {
    if (!trianglesArray || trianglesArray.length === 0) throw new Error('BVH: empty triangle list');
    const seeds = trianglesArray.map((t, i) => {
      const a = t.triangle.a, b = t.triangle.b, c = t.triangle.c;
      const min = [Math.min(a.x, b.x, c.x), Math.min(a.y, b.y, c.y), Math.min(a.z, b.z, c.z)];
      const max = [Math.max(a.x, b.x, c.x), Math.max(a.y, b.y, c.y), Math.max(a.z, b.z, c.z)];
      return { min, max, centroid: [(min[0] + max[0]) * 0.5, (min[1] + max[1]) * 0.5, (min[2] + max[2]) * 0.5], index: i };
    });
    this.mins = []; this.maxs = []; this.miss = []; this.first = []; this.count = [];
    this.inorderTrianglesIndicesArray = [];
    this._build(seeds, 0);
    this.nodeCount = this.miss.length;
    for (let i = 0; i < this.nodeCount; i++) if (this.miss[i] >= this.nodeCount) this.miss[i] = -1;
  }

  _build(list, depth) {
    const id = this.miss.length;
    this.mins.push(null); this.maxs.push(null); this.miss.push(0); this.first.push(-2); this.count.push(0);
    let min, max;
    if (list.length <= LEAF_SIZE) {
      min = [Infinity, Infinity, Infinity]; max = [-Infinity, -Infinity, -Infinity];
      this.first[id] = this.inorderTrianglesIndicesArray.length;
      this.count[id] = list.length;
      for (const s of list) {
        for (let k = 0; k < 3; k++) { min[k] = Math.min(min[k], s.min[k]); max[k] = Math.max(max[k], s.max[k]); }
        this.inorderTrianglesIndicesArray.push(s.index);
      }
    } else {
      const axis = depth % 3;
      list.sort((p, q) => p.centroid[axis] - q.centroid[axis]); // stable in V8 >= 7.0
      const mid = Math.floor(list.length / 2);
      const l = this._build(list.slice(0, mid), depth + 1);
      const r = this._build(list.slice(mid), depth + 1);
      min = [0, 1, 2].map((k) => Math.min(this.mins[l][k], this.mins[r][k]));
      max = [0, 1, 2].map((k) => Math.max(this.maxs[l][k], this.maxs[r][k]));
    }
    this.mins[id] = min; this.maxs[id] = max;
    this.miss[id] = this.miss.length; // first pre-order index after this subtree (fixed up to -1 at the end)
    return id;
  }

  /** The five node arrays of bvh.js:271-306 (3 numbers per node, ints triplicated) -- reference-shaped. */
  convertToTexturizableArrays() {
    const tri = (a) => { const o = []; a.forEach((v) => o.push(v, v, v)); return o; };
    const flat = (a) => { const o = []; a.forEach((v) => o.push(v[0], v[1], v[2])); return o; };
    return {
      nodesBoundingBoxesMins: flat(this.mins), nodesBoundingBoxesMaxs: flat(this.maxs),
      nodesTrianglesIndices: [], nodesTrianglesCount: tri(this.count),
      nodesInorderTrianglesIndices: tri(this.first), nodesMissLinkIndices: tri(this.miss),
    };
  }

  /** Typed, de-triplicated arrays: the layout ptx_upload_scene takes (include/ptx.h). */
  toTyped() {
    const N = this.nodeCount;
    const nodeMin = new Float32Array(3 * N), nodeMax = new Float32Array(3 * N);
    for (let i = 0; i < N; i++) for (let k = 0; k < 3; k++) { nodeMin[3 * i + k] = this.mins[i][k]; nodeMax[3 * i + k] = this.maxs[i][k]; }
    return {
      N, nodeMin, nodeMax, nodeMiss: Int32Array.from(this.miss), nodeFirst: Int32Array.from(this.first),
      nodeCount: Int32Array.from(this.count), inorderTri: Int32Array.from(this.inorderTrianglesIndicesArray),
    };
  }
}

module.exports = { BVH, LEAF_SIZE };
