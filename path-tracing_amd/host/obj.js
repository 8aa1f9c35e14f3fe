'use strict';
/*
 * Wavefront OBJ + MTL reader.  The reference has no live OBJ path (main.js:22 is commented out); this
 * exists for BASELINE.json config 3 (resources/teapot/teapot.obj) and returns the same object-tree
 * shape gltf.js does: one Mesh per (object, material) run, polygons fan-triangulated
 * (v0, v[i], v[i+1]), material { color: Kd, emissive: Ke }.
 */
const fs = require('fs');
const path = require('path');
const { makeObject } = require('./gltf');

function parseMTL(file) {
  const mats = {};
  let cur = null;
  for (const line of fs.readFileSync(file, 'utf8').split(/\r?\n/)) {
    const t = line.trim().split(/\s+/);
    if (t[0] === 'newmtl') { cur = { name: t[1], color: { r: 1, g: 1, b: 1 }, emissive: { r: 0, g: 0, b: 0 } }; mats[t[1]] = cur; }
    else if (cur && t[0] === 'Kd') cur.color = { r: +t[1], g: +t[2], b: +t[3] };
    else if (cur && t[0] === 'Ke') cur.emissive = { r: +t[1], g: +t[2], b: +t[3] };
  }
  return mats;
}

function loadOBJ(file) {
  const verts = [];
  let mats = {};
  const root = makeObject('Group', path.basename(file));
  let curMat = { name: '', color: { r: 1, g: 1, b: 1 }, emissive: { r: 0, g: 0, b: 0 } };
  let pos = null, name = '';
  const flush = () => {
    if (pos && pos.length) {
      const o = makeObject('Mesh', name);
      o.geometry = { position: Float32Array.from(pos), index: null, bboxMin: null, bboxMax: null };
      o.material = curMat;
      root.children.push(o);
    }
    pos = [];
  };
  flush();
  for (const line of fs.readFileSync(file, 'utf8').split(/\r?\n/)) {
    const t = line.trim().split(/\s+/);
    switch (t[0]) {
      case 'v': verts.push([+t[1], +t[2], +t[3]]); break;
      case 'mtllib': { const f = path.join(path.dirname(file), t.slice(1).join(' ')); if (fs.existsSync(f)) mats = parseMTL(f); break; }
      case 'usemtl': flush(); if (mats[t[1]]) curMat = mats[t[1]]; break;
      case 'o': case 'g': flush(); name = t.slice(1).join(' '); break;
      case 'f': {
        const idx = t.slice(1).map((s) => { const i = parseInt(s.split('/')[0], 10); return i < 0 ? verts.length + i : i - 1; });
        for (let i = 1; i + 1 < idx.length; i++) for (const k of [idx[0], idx[i], idx[i + 1]]) pos.push(verts[k][0], verts[k][1], verts[k][2]);
        break;
      }
      default: break;
    }
  }
  flush();
  return { scene: root };
}

module.exports = { loadOBJ };
