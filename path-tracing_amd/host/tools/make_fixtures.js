#!/usr/bin/env node
'use strict';
/*
 * Generates tests/golden/scenes/*.ptxs (+ .json side-cars) from the scene files under
 * /root/reference/resources with the build's own JS loaders.  Run in the build container only
 * (the reference tree does not travel to the GPU box):
 *     node path-tracing_amd/host/tools/make_fixtures.js [/root/reference/resources] [outdir]
 * The outputs are DATA (flattened geometry + BVH arrays + camera vectors), not reference source.
 */
const fs = require('fs');
const path = require('path');
const { loadModel, buildSceneArrays } = require('../scene');
const { writeScene } = require('../scenefile');
const { cameraFromPreset } = require('../camera');

const res = process.argv[2] || '/root/reference/resources';
const out = process.argv[3] || path.join(__dirname, '..', '..', '..', 'tests', 'golden', 'scenes');
fs.mkdirSync(out, { recursive: true });

function quad(y, hx, hz, down) {
  // two triangles spanning [-hx,hx] x [-hz,hz] at height y; `down` flips the winding so that the
  // geometric normal normalize((c-b) x (a-b)) points to -y
  const a = [-hx, y, -hz], b = [hx, y, -hz], c = [hx, y, hz], d = [-hx, y, hz];
  const t = down ? [a, b, c, a, c, d] : [a, c, b, a, d, c];
  return [].concat(...t);
}

/* BASELINE.json configs (SURVEY.md 8d).  Overrides are the documented ones for C1 / C3. */
const SCENES = [
  { name: 'cornell_faithful', file: 'cornell/glTF/scene.gltf', camera: 'cornell-box', note: 'C1 geometry, faithful loader: L = 0 (glTF dropped the OBJ Ke), image is black' },
  { name: 'cornell', file: 'cornell/glTF/scene.gltf', camera: 'cornell-box', opts: { materialOverrides: { light: { emissive: [1, 1, 1] } } },
    note: 'C1 with the documented override: material "light" gets emissive (1,1,1)' },
  { name: 'my_cornell_6', file: 'my_cornell_6/gltf/my_cornell_6.gltf', camera: 'cornell', note: 'C2' },
  { name: 'teapot', file: 'teapot/teapot.obj', camera: 'teapot-obj',
    opts: { extraMeshes: [
      { name: 'floor', positions: quad(-39.5033, 400, 400, false), color: [0.7, 0.7, 0.7], emissive: [0, 0, 0] },
      { name: 'light', positions: quad(160, 60, 60, true), color: [1, 1, 1], emissive: [1, 1, 1] }] },
    note: 'C3 with the documented override: floor quad + 2-triangle emitter (teapot.obj has Ke 0)' },
  { name: 'bedroom2_v5', file: 'bedroom2/gltf/v5/bedroom2_v5.gltf', camera: 'room-v3', note: 'C4' },
  { name: 'pixar_room', file: 'pixar_room/v1/pixar-room.gltf', camera: 'pixar-room', note: 'C5 + headline' },
];

const dumpF64 = new Set(['cornell', 'my_cornell_6']); // small scenes also keep their f64 triangles for the BVH oracle

for (const sc of SCENES) {
  const t0 = Date.now();
  const { trianglesArray } = loadModel(path.join(res, sc.file), sc.opts);
  const arrays = buildSceneArrays(trianglesArray);
  writeScene(path.join(out, sc.name + '.ptxs'), arrays);
  const cam = cameraFromPreset(sc.camera, 1, 1).getCamera();
  const v = (p) => [p.x, p.y, p.z];
  const side = { name: sc.name, source: 'resources/' + sc.file, note: sc.note, T: arrays.T, N: arrays.N, L: arrays.L, camera_preset: sc.camera,
    camera: { cam_src: v(cam.cameraSource), cam_dir: v(cam.cameraDirection), cam_up: v(cam.cameraUp), cam_right: v(cam.cameraRight), cam_left_bottom: v(cam.cameraLeftBottom) } };
  if (dumpF64.has(sc.name)) {
    side.triangles_f64 = trianglesArray.map((t) => [t.triangle.a.x, t.triangle.a.y, t.triangle.a.z, t.triangle.b.x, t.triangle.b.y, t.triangle.b.z, t.triangle.c.x, t.triangle.c.y, t.triangle.c.z]);
  }
  fs.writeFileSync(path.join(out, sc.name + '.json'), JSON.stringify(side, null, 1));
  console.log(sc.name, 'T', arrays.T, 'N', arrays.N, 'L', arrays.L, (Date.now() - t0) + 'ms');
}
