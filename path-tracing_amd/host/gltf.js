'use strict';
/*
 * glTF 2.0 (.gltf + .bin / data: URIs / .glb) reader producing the object tree three's GLTFLoader
 * would hand to /root/reference/main.js:545 -- only the parts main.js:551-679 looks at: the node
 * hierarchy with TRS, Mesh objects with a float POSITION attribute (+ optional index) and a material
 * reduced to { color, emissive }.
 *
 * three r163 behaviour restated (not importable offline; parity unpinned, SURVEY.md 8c-3):
 *  - scene root = Group; children in scene.nodes order; a node's mesh object comes before the node's
 *    own children; a mesh with several primitives is a Group of Meshes in primitive order;
 *  - node.matrix is applied with Object3D.applyMatrix4 (premultiply then decompose into TRS), TRS
 *    fields are copied as written (quaternion NOT normalised);
 *  - material.color = baseColorFactor.rgb (linear, default 1,1,1); material.emissive =
 *    emissiveFactor (default 0,0,0); KHR_materials_emissive_strength only sets emissiveIntensity,
 *    which main.js:667 never reads; a primitive without material gets white / black;
 *  - geometry.boundingBox = accessor min/max of POSITION when present, else computed from the data.
 */
const fs = require('fs');
const path = require('path');
const { mat4 } = require('./math3');

const COMPONENT = { 5120: [Int8Array, 1], 5121: [Uint8Array, 1], 5122: [Int16Array, 2], 5123: [Uint16Array, 2], 5125: [Uint32Array, 4], 5126: [Float32Array, 4] };
const NCOMP = { SCALAR: 1, VEC2: 2, VEC3: 3, VEC4: 4, MAT2: 4, MAT3: 9, MAT4: 16 };

function parseContainer(file) {
  const raw = fs.readFileSync(file);
  if (raw.length >= 12 && raw.readUInt32LE(0) === 0x46546c67) { // 'glTF' binary container
    let off = 12, json = null, bin = null;
    while (off < raw.length) {
      const len = raw.readUInt32LE(off), type = raw.readUInt32LE(off + 4);
      const chunk = raw.slice(off + 8, off + 8 + len);
      if (type === 0x4e4f534a) json = JSON.parse(chunk.toString('utf8'));
      else if (type === 0x004e4942) bin = chunk;
      off += 8 + len;
    }
    return { json, glbBin: bin };
  }
  return { json: JSON.parse(raw.toString('utf8')), glbBin: null };
}

function loadBuffers(json, dir, glbBin) {
  return (json.buffers || []).map((b, i) => {
    if (b.uri === undefined) {
      if (i === 0 && glbBin) return glbBin;
      throw new Error('glTF: buffer ' + i + ' has no uri');
    }
    if (b.uri.startsWith('data:')) return Buffer.from(b.uri.slice(b.uri.indexOf(',') + 1), 'base64');
    return fs.readFileSync(path.join(dir, decodeURIComponent(b.uri)));
  });
}

/** Reads accessor `idx` into a flat typed array (tightly packed), honouring byteStride. */
function readAccessor(json, buffers, idx) {
  const acc = json.accessors[idx];
  if (acc.sparse) throw new Error('glTF: sparse accessors are not supported');
  const [Typed, csize] = COMPONENT[acc.componentType];
  const nc = NCOMP[acc.type];
  const out = new Typed(acc.count * nc);
  if (acc.bufferView === undefined) return out;
  const bv = json.bufferViews[acc.bufferView];
  const buf = buffers[bv.buffer];
  const base = (bv.byteOffset || 0) + (acc.byteOffset || 0);
  const stride = bv.byteStride || csize * nc;
  const dv = new DataView(buf.buffer, buf.byteOffset, buf.byteLength);
  const get = { 5120: 'getInt8', 5121: 'getUint8', 5122: 'getInt16', 5123: 'getUint16', 5125: 'getUint32', 5126: 'getFloat32' }[acc.componentType];
  for (let i = 0; i < acc.count; i++) for (let c = 0; c < nc; c++) out[i * nc + c] = dv[get](base + i * stride + c * csize, true);
  return out;
}

function makeMaterial(json, idx) {
  if (idx === undefined) return { name: '', color: { r: 1, g: 1, b: 1 }, emissive: { r: 0, g: 0, b: 0 } };
  const m = json.materials[idx];
  const pbr = m.pbrMetallicRoughness || {};
  const c = pbr.baseColorFactor || [1, 1, 1, 1];
  const e = m.emissiveFactor || [0, 0, 0];
  return { name: m.name || '', color: { r: c[0], g: c[1], b: c[2] }, emissive: { r: e[0], g: e[1], b: e[2] } };
}

/** TRIANGLE_STRIP / TRIANGLE_FAN -> triangle list (BufferGeometryUtils.toTrianglesDrawMode order). */
function toTriangleList(index, count, mode) {
  const at = (i) => (index ? index[i] : i);
  const n = index ? index.length : count;
  const out = [];
  if (mode === 6) for (let i = 1; i < n - 1; i++) out.push(at(0), at(i), at(i + 1));
  else for (let i = 0; i < n - 2; i++) { if (i % 2 === 0) out.push(at(i), at(i + 1), at(i + 2)); else out.push(at(i + 2), at(i + 1), at(i)); }
  return Uint32Array.from(out);
}

function makeObject(kind, name) {
  return { kind, name: name || '', visible: true, position: [0, 0, 0], quaternion: [0, 0, 0, 1], scale: [1, 1, 1], children: [], matrix: mat4.identity(), matrixWorld: mat4.identity() };
}

function loadGLTF(file) {
  const { json, glbBin } = parseContainer(file);
  const buffers = loadBuffers(json, path.dirname(file), glbBin);

  const meshObjects = (meshIdx) => {
    const def = json.meshes[meshIdx];
    const meshes = [];
    for (const prim of def.primitives) {
      const mode = prim.mode === undefined ? 4 : prim.mode;
      if (mode !== 4 && mode !== 5 && mode !== 6) continue; // points / lines are not THREE.Mesh
      const pa = json.accessors[prim.attributes.POSITION];
      if (pa.componentType !== 5126 || pa.type !== 'VEC3') throw new Error('glTF: POSITION must be float VEC3');
      const position = readAccessor(json, buffers, prim.attributes.POSITION);
      let index = prim.indices !== undefined ? readAccessor(json, buffers, prim.indices) : null;
      if (mode !== 4) index = toTriangleList(index, pa.count, mode);
      const o = makeObject('Mesh', def.name);
      o.geometry = { position, index, bboxMin: pa.min ? pa.min.slice(0, 3) : null, bboxMax: pa.max ? pa.max.slice(0, 3) : null };
      o.material = makeMaterial(json, prim.material);
      meshes.push(o);
    }
    if (meshes.length === 1) return meshes[0];
    const g = makeObject('Group', def.name);
    g.children.push(...meshes);
    return g;
  };

  const buildNode = (nodeIdx) => {
    const def = json.nodes[nodeIdx];
    const o = def.mesh !== undefined ? meshObjects(def.mesh) : makeObject('Object3D', def.name);
    o.name = def.name || o.name;
    if (def.matrix !== undefined) {
      const d = mat4.decompose(mat4.multiply(def.matrix.slice(), mat4.identity()));
      o.position = d.position; o.quaternion = d.quaternion; o.scale = d.scale;
    } else {
      if (def.translation) o.position = def.translation.slice();
      if (def.rotation) o.quaternion = def.rotation.slice();
      if (def.scale) o.scale = def.scale.slice();
    }
    for (const c of def.children || []) o.children.push(buildNode(c));
    return o;
  };

  const sceneIdx = json.scene !== undefined ? json.scene : 0;
  const root = makeObject('Group', (json.scenes[sceneIdx] || {}).name);
  for (const n of json.scenes[sceneIdx].nodes || []) root.children.push(buildNode(n));
  return { scene: root, json };
}

module.exports = { loadGLTF, readAccessor, makeObject };
