#!/usr/bin/env node
'use strict';
/* Rebuilds the BVH arrays of a .ptxs file from its (f32) triangle coordinates with host/bvh.js and writes
 * a new .ptxs.  Used by tests to cross-check the JS builder against oracle/bvh_oracle.py on the same input. */
const { readScene, writeScene } = require('../scenefile');
const { BVH } = require('../bvh');
const [, , inFile, outFile] = process.argv;
if (!inFile || !outFile) { console.error('usage: rebuild_bvh.js in.ptxs out.ptxs'); process.exit(2); }
const s = readScene(inFile);
const tris = [];
for (let i = 0; i < s.T; i++) {
  const c = s.coords;
  tris.push({ triangle: { a: { x: c[9 * i], y: c[9 * i + 1], z: c[9 * i + 2] }, b: { x: c[9 * i + 3], y: c[9 * i + 4], z: c[9 * i + 5] }, c: { x: c[9 * i + 6], y: c[9 * i + 7], z: c[9 * i + 8] } } });
}
const b = new BVH(tris).toTyped();
writeScene(outFile, Object.assign({}, s, b));
