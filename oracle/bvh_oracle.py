"""CPU ORACLE for the host-side BVH builder (TEST INFRASTRUCTURE ONLY).

Literal Python restatement of /root/reference/classes/bvh.js:8-306 + classes/bvhNode.js:4-25, including the parts
the product's builder (path-tracing_amd/host/bvh.js) replaces by a direct computation: the global nodeId counter,
the BFS of addMissLinks (bvh.js:161-187) and the id -> pre-order index mapping of updateMissLinks (bvh.js:233-239,
done with a dict instead of map+indexOf).  All arithmetic in f64 like the reference.

Parity pin: the reference's builder needs three.js (not vendored, not installable) so it cannot run here; this
restatement is pinned by the BVH shapes SURVEY.md 8a records from a survey-time run of the reference builder
(node counts, leaf-size histograms, the cornell miss/first/count table).  Leaf contents: parity unpinned.
"""


class _Node:
    next_id = 0  # bvhNode.js:5 static nodeId

    def __init__(self, bmin, bmax, left=None, right=None, tri_idx=None):
        self.node_id = _Node.next_id  # bvhNode.js:11
        _Node.next_id += 1
        self.bmin, self.bmax = bmin, bmax
        self.left, self.right = left, right
        self.tri_idx = tri_idx if tri_idx is not None else [-1]
        self.centroid = [(bmin[k] + bmax[k]) * 0.5 for k in range(3)]  # Box3.getCenter
        self.miss = None
        self.first = None
        self.count = 0


def _build(nodes, depth, n_leaf, counter):  # bvh.js:74-118
    if len(nodes) <= n_leaf:
        bmin = [float("inf")] * 3
        bmax = [float("-inf")] * 3
        idx = []
        for t in nodes:
            bmin = [min(bmin[k], t.bmin[k]) for k in range(3)]
            bmax = [max(bmax[k], t.bmax[k]) for k in range(3)]
            idx.extend(t.tri_idx)
        counter[0] += 1
        return _Node(bmin, bmax, None, None, idx)
    axis = depth % 3
    nodes.sort(key=lambda t: t.centroid[axis])  # stable, like V8's TimSort; comparator a-b == ascending
    mid = len(nodes) // 2
    left = _build(nodes[:mid], depth + 1, n_leaf, counter)
    right = _build(nodes[mid:], depth + 1, n_leaf, counter)
    bmin = [min(left.bmin[k], right.bmin[k]) for k in range(3)]
    bmax = [max(left.bmax[k], right.bmax[k]) for k in range(3)]
    counter[0] += 1
    return _Node(bmin, bmax, left, right, [-2])


def build_bvh(triangles_f64):
    """triangles_f64: list of 9 floats (a.xyz b.xyz c.xyz). Returns dict of plain lists (de-triplicated)."""
    _Node.next_id = 0
    seeds = []
    for i, t in enumerate(triangles_f64):  # bvh.js:19-31
        pts = [t[0:3], t[3:6], t[6:9]]
        bmin = [min(p[k] for p in pts) for k in range(3)]
        bmax = [max(p[k] for p in pts) for k in range(3)]
        seeds.append(_Node(bmin, bmax, None, None, [i]))
    counter = [0]
    root = _build(seeds, 0, 5, counter)  # bvh.js:40

    # addMissLinks, bvh.js:161-187 (BFS with node ids)
    queue = [(root, -1)]
    while queue:
        node, parent_miss = queue.pop(0)
        if node is root:
            node.miss = -1
        elif queue and queue[0][1] == parent_miss:
            node.miss = queue[0][0].node_id
        else:
            node.miss = parent_miss
        if node.left is not None:
            queue.append((node.left, node.right.node_id if node.right else -1))
        if node.right is not None:
            queue.append((node.right, parent_miss))

    pre = []  # fillPreorderArray, bvh.js:149-159

    def walk(n):
        if n is None:
            return
        pre.append(n)
        walk(n.left)
        walk(n.right)

    walk(root)
    pos = {n.node_id: i for i, n in enumerate(pre)}  # updateMissLinks, bvh.js:233-239
    for n in pre:
        n.miss = pos.get(n.miss, -1)

    inorder = []  # addDoublePointer, bvh.js:241-269
    for n in pre:
        if n.tri_idx[0] >= 0:
            inorder.extend(n.tri_idx)
    where = {}
    for i, v in enumerate(inorder):
        where.setdefault(v, i)
    for n in pre:
        if n.tri_idx[0] != -2:
            n.first = where[n.tri_idx[0]]
            n.count = len(n.tri_idx)
        else:
            n.first = -1
            n.count = 0
    return {  # convertToTexturizableArrays, bvh.js:271-306 (de-triplicated)
        "node_count": counter[0],
        "node_min": [n.bmin for n in pre], "node_max": [n.bmax for n in pre],
        "node_miss": [n.miss for n in pre],
        "node_first": [n.first if n.tri_idx[0] >= 0 else n.tri_idx[0] for n in pre],
        "node_cnt": [n.count if n.count else 0 for n in pre],
        "inorder_tri": inorder,
    }
