'use strict';
/*
 * Scene ingestion: host-side counterpart of loadModel() in /root/reference/main.js:536-712 and of
 * utils/triangleMapper.js:43-79, without three.js.
 *
 *   loadModel(file, opts) -> { trianglesArray }   (same per-triangle records main.js builds:
 *        { triangle:{a,b,c}, normal, color, emission })
 *   mapTrianglesArrayToTexturizedArray(trianglesArray) -> { coordinates, normals, colors, emissions,
 *        lightIndices }  (reference-shaped plain arrays, ints triplicated)
 *   buildSceneArrays(trianglesArray) -> typed, de-triplicated arrays for ptx_upload_scene
 *
 * Steps reproduced from main.js: updateMatrixWorld (551); recentre by minus the centre of
 * Box3.setFromObject (553-568: non-precise = each mesh's local AABB pushed through matrixWorld by its
 * 8 corners; the translation is added to each Mesh's LOCAL position); re-derive each mesh's world
 * matrix (571-576); traverseVisible in depth-first pre-order (579); toNonIndexed (594);
 * applyMatrix4(matrixWorld) per vertex (606-614); one triangle per 3 vertices (618-631);
 * normal = THREE.Triangle.getNormal = normalize((c-b) x (a-b)) or 0 (652-653); color = material.color
 * (660); emission = material.emissive (667).  Everything is f64, like the reference.
 */
const path = require('path');
const { v3, mat4 } = require('./math3');
const { loadGLTF, makeObject } = require('./gltf');
const { loadOBJ } = require('./obj');
const { BVH } = require('./bvh');

const EMISSION_CONSTANT = 50; // utils/triangleMapper.js:44

function traverse(o, fn) { fn(o); for (const c of o.children) traverse(c, fn); }
function traverseVisible(o, fn) { if (!o.visible) return; fn(o); for (const c of o.children) traverseVisible(c, fn); }

function updateMatrixWorld(o, parentWorld) {
  o.matrix = mat4.compose(o.position, o.quaternion, o.scale);
  o.matrixWorld = parentWorld ? mat4.multiply(parentWorld, o.matrix) : o.matrix.slice();
  for (const c of o.children) updateMatrixWorld(c, o.matrixWorld);
}

function localBox(g) {
  if (g.bboxMin && g.bboxMax) return { min: g.bboxMin, max: g.bboxMax };
  const min = [Infinity, Infinity, Infinity], max = [-Infinity, -Infinity, -Infinity];
  for (let i = 0; i < g.position.length; i += 3) for (let k = 0; k < 3; k++) { min[k] = Math.min(min[k], g.position[i + k]); max[k] = Math.max(max[k], g.position[i + k]); }
  return { min, max };
}

function sceneCentre(root) {
  let min = [Infinity, Infinity, Infinity], max = [-Infinity, -Infinity, -Infinity];
  traverse(root, (o) => {
    if (o.kind !== 'Mesh') return;
    const b = localBox(o.geometry);
    for (let c = 0; c < 8; c++) {
      const p = v3.applyMatrix4([c & 1 ? b.max[0] : b.min[0], c & 2 ? b.max[1] : b.min[1], c & 4 ? b.max[2] : b.min[2]], o.matrixWorld);
      min = v3.min(min, p); max = v3.max(max, p);
    }
  });
  return v3.scale(v3.add(min, max), 0.5);
}

/**
 * opts.materialOverrides: { [materialName]: { color:[r,g,b], emissive:[r,g,b] } } applied before flattening
 * opts.extraMeshes: [{ name, positions:[...9 per triangle, world space AFTER recentring], color, emissive }]
 * Both exist only for the documented C1/C3 overrides (SURVEY.md 8d); default = faithful loader.
 */
function loadModel(file, opts) {
  opts = opts || {};
  const ext = path.extname(file).toLowerCase();
  const model = ext === '.obj' ? loadOBJ(file) : loadGLTF(file);
  const root = model.scene;

  updateMatrixWorld(root, null);
  const centre = sceneCentre(root);
  const translation = [-centre[0], -centre[1], -centre[2]];
  traverse(root, (o) => { if (o.kind === 'Mesh') o.position = v3.add(o.position, translation); });
  // main.js:571-576: each mesh recomposes its local matrix and re-derives matrixWorld from the parent's
  // (unchanged) world matrix; descendants of a mesh are refreshed too (updateMatrixWorld(true)).
  const refresh = (o, parentWorld) => {
    if (o.kind === 'Mesh') updateMatrixWorld(o, parentWorld);
    else for (const c of o.children) refresh(c, o.matrixWorld);
  };
  refresh(root, null);

  const trianglesArray = [];
  traverseVisible(root, (o) => {
    if (o.kind !== 'Mesh') return;
    const g = o.geometry;
    let mat = o.material;
    const ov = opts.materialOverrides && opts.materialOverrides[mat.name];
    if (ov) {
      mat = { name: mat.name, color: ov.color ? { r: ov.color[0], g: ov.color[1], b: ov.color[2] } : mat.color,
              emissive: ov.emissive ? { r: ov.emissive[0], g: ov.emissive[1], b: ov.emissive[2] } : mat.emissive };
    }
    const n = g.index ? g.index.length : g.position.length / 3;
    const w = new Array(n);
    for (let i = 0; i < n; i++) {
      const k = g.index ? g.index[i] : i;
      w[i] = v3.applyMatrix4([g.position[3 * k], g.position[3 * k + 1], g.position[3 * k + 2]], o.matrixWorld);
    }
    for (let t = 0; t + 2 < n; t += 3) trianglesArray.push(makeTriangle(w[t], w[t + 1], w[t + 2], mat.color, mat.emissive));
  });
  for (const m of opts.extraMeshes || []) {
    const col = { r: m.color[0], g: m.color[1], b: m.color[2] }, em = { r: m.emissive[0], g: m.emissive[1], b: m.emissive[2] };
    for (let i = 0; i + 8 < m.positions.length; i += 9) {
      const p = m.positions;
      trianglesArray.push(makeTriangle([p[i], p[i + 1], p[i + 2]], [p[i + 3], p[i + 4], p[i + 5]], [p[i + 6], p[i + 7], p[i + 8]], col, em));
    }
  }
  return { trianglesArray, centre };
}

function makeTriangle(a, b, c, color, emission) {
  // THREE.Triangle.getNormal
  let n = v3.cross(v3.sub(c, b), v3.sub(a, b));
  const l2 = v3.dot(n, n);
  n = l2 > 0 ? v3.scale(n, 1 / Math.sqrt(l2)) : [0, 0, 0];
  const rec = (p) => ({ x: p[0], y: p[1], z: p[2] });
  return { triangle: { a: rec(a), b: rec(b), c: rec(c) }, normal: rec(n), color, emission };
}

/** utils/triangleMapper.js:43-79, reference-shaped output. */
function mapTrianglesArrayToTexturizedArray(trianglesArray) {
  const coordinates = [], normals = [], colors = [], emissions = [], lightIndices = [];
  trianglesArray.forEach((t, i) => {
    const a = t.triangle.a, b = t.triangle.b, c = t.triangle.c;
    coordinates.push(a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z);
    normals.push(t.normal.x, t.normal.y, t.normal.z);
    colors.push(t.color.r, t.color.g, t.color.b);
    emissions.push(t.emission.r * EMISSION_CONSTANT, t.emission.g * EMISSION_CONSTANT, t.emission.b * EMISSION_CONSTANT);
    if (t.emission.r > 0 || t.emission.g > 0 || t.emission.b > 0) lightIndices.push(i, i, i);
  });
  return { coordinates, normals, colors, emissions, lightIndices };
}

/** Everything ptx_upload_scene needs, typed and de-triplicated (f64 -> f32 happens here, once). */
function buildSceneArrays(trianglesArray) {
  const m = mapTrianglesArrayToTexturizedArray(trianglesArray);
  const bvh = new BVH(trianglesArray);
  const lights = [];
  for (let i = 0; i < m.lightIndices.length; i += 3) lights.push(m.lightIndices[i]);
  return Object.assign({
    T: trianglesArray.length, L: lights.length,
    coords: Float32Array.from(m.coordinates), normals: Float32Array.from(m.normals),
    colors: Float32Array.from(m.colors), emissions: Float32Array.from(m.emissions),
    lightTri: Int32Array.from(lights),
  }, bvh.toTyped());
}

module.exports = { loadModel, mapTrianglesArrayToTexturizedArray, buildSceneArrays, EMISSION_CONSTANT };
