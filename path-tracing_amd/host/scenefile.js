'use strict';
/*
 * .ptxs scene container: the eleven scene arrays of ptx_upload_scene in one little-endian file so the
 * Python/C side of the repo (tests, bench.py) can load what the JS host produced.
 *   u32 magic 'PTXS', u32 version (1), u32 T, u32 N, u32 L, u32 reserved[3]
 *   f32 coords[9T] normals[3T] colors[3T] emissions[3T]; i32 light_tri[L];
 *   f32 node_min[3N] node_max[3N]; i32 node_miss[N] node_first[N] node_count[N]; i32 inorder_tri[T]
 */
const fs = require('fs');
const MAGIC = 0x53585450; // 'PTXS'
const ORDER = [['coords', 'f', 9, 'T'], ['normals', 'f', 3, 'T'], ['colors', 'f', 3, 'T'], ['emissions', 'f', 3, 'T'], ['lightTri', 'i', 1, 'L'],
  ['nodeMin', 'f', 3, 'N'], ['nodeMax', 'f', 3, 'N'], ['nodeMiss', 'i', 1, 'N'], ['nodeFirst', 'i', 1, 'N'], ['nodeCount', 'i', 1, 'N'], ['inorderTri', 'i', 1, 'T']];

function writeScene(file, s) {
  const head = Buffer.alloc(32);
  head.writeUInt32LE(MAGIC, 0); head.writeUInt32LE(1, 4); head.writeUInt32LE(s.T, 8); head.writeUInt32LE(s.N, 12); head.writeUInt32LE(s.L, 16);
  const parts = [head];
  for (const [name, , per, dim] of ORDER) {
    const a = s[name];
    if (a.length !== per * s[dim]) throw new Error('writeScene: ' + name + ' has ' + a.length + ' elements, expected ' + per * s[dim]);
    parts.push(Buffer.from(a.buffer, a.byteOffset, a.byteLength));
  }
  fs.writeFileSync(file, Buffer.concat(parts));
}

function readScene(file) {
  const b = fs.readFileSync(file);
  if (b.readUInt32LE(0) !== MAGIC || b.readUInt32LE(4) !== 1) throw new Error('readScene: bad header');
  const s = { T: b.readUInt32LE(8), N: b.readUInt32LE(12), L: b.readUInt32LE(16) };
  let off = 32;
  for (const [name, kind, per, dim] of ORDER) {
    const n = per * s[dim];
    const ab = b.buffer.slice(b.byteOffset + off, b.byteOffset + off + 4 * n);
    s[name] = kind === 'f' ? new Float32Array(ab) : new Int32Array(ab);
    off += 4 * n;
  }
  return s;
}

module.exports = { writeScene, readScene };
