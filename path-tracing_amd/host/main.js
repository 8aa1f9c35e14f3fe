#!/usr/bin/env node
'use strict';
/*
 * main.js -- Node CLI counterpart of /root/reference/main.js (the browser page's top-level module):
 * load a scene (main.js:78-90, 536-712), set up the camera preset (main.js:144-176), build + flatten the BVH
 * (main.js:180-190), upload once, render `frames` frames of `sampleCount` paths per pixel (main.js:391-438),
 * read the image back and save it (main.js:454-499).  What changed on purpose: source-level constants
 * (main.js:1-15) became flags, the scene is uploaded once instead of every frame, there is no 32x32 tiling and no
 * sleeping (both existed to dodge the browser's GPU watchdog), `timestamp` is a fixed list (1000 + frame) instead
 * of performance.now() so renders are reproducible, and frames accumulate in f32 instead of RGBA8.
 *
 *   node main.js --scene file.gltf|file.obj|file.ptxs [--camera preset | --camera-json file] [--width 256]
 *        [--height 256] [--frames 5] [--spp 5] [--bounces 5] [--out image.png|.ppm] [--dump-sum file.f32]
 *        [--device 0 | --gpus N | --devices 0,1,...] [--traversal 0..3] [--prune 1|2] [--margin-lat ppm] [--save-scene out.ptxs] [--light-material name]
 *        [--checkpoint file.ptxc] [--checkpoint-every 16] [--compat-out image.png [--compat-rules 1|2] [--compat-gamma 0|1]]
 * --gpus N (devices 0 .. N-1) or --devices a,b,...: one PathTracer over several GPUs.  Frame f of every renderFrames call
 *   goes to device f mod N over the full image, the devices render side by side, and the per-device float buffers are
 *   summed by one RCCL all-reduce when the image (or a checkpoint) is read -- the reference's frame loop (main.js:391-438)
 *   for a whole node.  The same device listed N times rehearses that path on one GPU (summed by a kernel, no RCCL).
 * --prune: rule for skipping BVH nodes behind the best hit (include/ptx.h): 2 = margin rule with lateral allowance (default,
 *   fastest; equal to the unpruned walk on every frame of every BASELINE job, but not proven), 1 = rule with a proof
 *   (DESIGN.md section 4).
 * --checkpoint: progressive rendering like renderAsync (main.js:391-436): frames are rendered in batches of
 *   --checkpoint-every; after each batch {sum buffer, frames done} is written to the file (atomically, via rename).  If
 *   the file exists when the program starts, rendering resumes after the frames it holds (same size required); the
 *   timestamp of frame f is always 1000 + f, so a resumed render equals an uninterrupted one bit for bit.
 * --compat-out: also keep the reference's own RGBA8 ping-pong blend (glsl:532-552) and save it, flipped like main.js:464-473.
 *   --compat-gamma 1 passes it through the display pass first (fragmentShaderOutput.glsl): the image the reference saves.
 *   --compat-rules: what GL leaves to the implementation (8-bit store, NEAREST on texel boundaries): 1 = preferred rules
 *   (default), 2 = SwiftShader's, byte-identical to the reference run on SwiftShader (include/ptx.h, DESIGN.md section 4).
 */
const fs = require('fs');
const path = require('path');
const { loadModel, buildSceneArrays } = require('./scene');
const { readScene, writeScene } = require('./scenefile');
const { cameraFromPreset } = require('./camera');
const { PathTracer } = require('./ptx');
const { writePNG, writePPM } = require('./png');

function parseArgs(argv) {
  const a = { camera: 'teapot', width: 256, height: 256, frames: 5, spp: 5, bounces: 5, device: 0 }; // defaults = main.js:7-10,164-167
  for (let i = 2; i < argv.length; i++) {
    const k = argv[i].replace(/^--/, '').replace(/-([a-z])/g, (m, c) => c.toUpperCase());
    const v = argv[++i];
    if (v === undefined) throw new Error('missing value for ' + argv[i - 1]);
    a[k] = /^-?\d+$/.test(v) ? parseInt(v, 10) : v;
  }
  if (!a.scene) throw new Error('usage: node main.js --scene <file> [options]  (see the header of main.js)');
  return a;
}

function main() {
  const a = parseArgs(process.argv);
  const t0 = Date.now();
  let scene;
  if (path.extname(a.scene) === '.ptxs') scene = readScene(a.scene);
  else {
    const opts = a.lightMaterial ? { materialOverrides: { [a.lightMaterial]: { emissive: [1, 1, 1] } } } : {};
    scene = buildSceneArrays(loadModel(a.scene, opts).trianglesArray);
  }
  if (a.saveScene) writeScene(a.saveScene, scene);
  console.log(`scene: T=${scene.T} N=${scene.N} L=${scene.L} (${Date.now() - t0} ms)`);

  let camera;
  if (a.cameraJson) {
    const c = JSON.parse(fs.readFileSync(a.cameraJson, 'utf8')).camera;
    camera = { cameraSource: c.cam_src, cameraUp: c.cam_up, cameraRight: c.cam_right, cameraLeftBottom: c.cam_left_bottom };
  } else camera = cameraFromPreset(a.camera, a.width, a.height).getCamera();

  let devices = a.device;
  if (a.devices !== undefined) devices = String(a.devices).split(',').map((d) => parseInt(d, 10));
  else if (a.gpus !== undefined && a.gpus > 1) devices = Array.from({ length: a.gpus }, (_, g) => g);
  if (Array.isArray(devices) && devices.length === 1) devices = devices[0];
  if (Array.isArray(devices) && a.compatOut) throw new Error('--compat-out needs a single device (the 8-bit blend is sequential over frames)');
  const pt = new PathTracer(devices);
  if (Array.isArray(devices)) console.log('devices: ' + devices.join(',') + ' (all-reduce over ' + pt.commWorld + ' ranks)');
  pt.uploadScene(scene);
  pt.setParams(a.width, a.height, camera, a.bounces, a.spp);
  if (a.traversal !== undefined) pt.setOption('traversal', a.traversal);
  if (a.prune !== undefined) pt.setOption('prune', a.prune);
  if (a.marginLat !== undefined) pt.setOption('margin_lat', a.marginLat);
  if (a.compatOut) {
    if (a.checkpoint) throw new Error('--compat-out cannot be combined with --checkpoint (the 8-bit target is not checkpointed)');
    const rules = a.compatRules === undefined ? 1 : a.compatRules;
    if (rules !== 1 && rules !== 2) throw new Error('--compat-rules must be 1 or 2');
    pt.setOption('compat_rgba8', rules);
  }
  let first = 0;
  if (a.checkpoint && fs.existsSync(a.checkpoint)) {
    const ck = readCheckpoint(a.checkpoint);
    if (ck.width !== a.width || ck.height !== a.height) throw new Error('checkpoint is ' + ck.width + 'x' + ck.height + ', not ' + a.width + 'x' + a.height);
    pt.writeSum(ck.sum, ck.framesDone);
    first = ck.framesDone;
    console.log('resumed ' + a.checkpoint + ' at frame ' + first);
  }
  const every = a.checkpoint ? Math.max(1, a.checkpointEvery || 16) : Math.max(1, a.frames);
  const t1 = Date.now();
  for (let f0 = first; f0 < a.frames; f0 += every) {
    const timestamps = [];
    for (let f = f0; f < Math.min(a.frames, f0 + every); f++) timestamps.push(1000 + f);
    pt.renderFrames(timestamps, f0);
    if (a.checkpoint) writeCheckpoint(a.checkpoint, a.width, a.height, pt.framesDone, pt.readSum());
  }
  const dt = (Date.now() - t1) / 1000;
  const c = pt.counters();
  console.log(`rendered ${a.frames} frames x ${a.spp} spp at ${a.width}x${a.height}, ${a.bounces} bounces in ${dt.toFixed(3)} s: ` +
    `${(c.samples / dt / 1e6).toFixed(2)} Msamples/s, ${(c.rays / dt / 1e6).toFixed(1)} Mrays/s (kernel ${c.kernelMs.toFixed(1)} ms)`);
  if (a.dumpSum) { const s = pt.readSum(); fs.writeFileSync(a.dumpSum, Buffer.from(s.buffer, s.byteOffset, s.byteLength)); }
  if (a.out) {
    const px = pt.readRgba8(true, true); // display gamma + vertical flip, like the output pass + main.js:464-473
    (path.extname(a.out) === '.ppm' ? writePPM : writePNG)(a.out, px, a.width, a.height);
    console.log('wrote ' + a.out);
  }
  if (a.compatOut) {
    const px = pt.readCompatRgba8(a.compatGamma === 1, true);
    (path.extname(a.compatOut) === '.ppm' ? writePPM : writePNG)(a.compatOut, px, a.width, a.height);
    console.log('wrote ' + a.compatOut);
  }
  pt.destroy();
}

// checkpoint file: 'PTXC' u32 version(1) u32 width u32 height u32 framesDone, then width*height*3 f32 (the raw sum)
function writeCheckpoint(file, width, height, framesDone, sum) {
  const head = Buffer.alloc(20);
  head.write('PTXC', 0, 'latin1');
  head.writeUInt32LE(1, 4); head.writeUInt32LE(width, 8); head.writeUInt32LE(height, 12); head.writeUInt32LE(framesDone, 16);
  fs.writeFileSync(file + '.tmp', Buffer.concat([head, Buffer.from(sum.buffer, sum.byteOffset, sum.byteLength)]));
  fs.renameSync(file + '.tmp', file);
}

function readCheckpoint(file) {
  const b = fs.readFileSync(file);
  if (b.length < 20 || b.toString('latin1', 0, 4) !== 'PTXC' || b.readUInt32LE(4) !== 1) throw new Error(file + ' is not a PTXC v1 checkpoint');
  const width = b.readUInt32LE(8), height = b.readUInt32LE(12), framesDone = b.readUInt32LE(16);
  if (b.length !== 20 + width * height * 12) throw new Error(file + ': truncated checkpoint');
  const sum = new Float32Array(width * height * 3);
  Buffer.from(sum.buffer).set(b.subarray(20));
  return { width, height, framesDone, sum };
}

try { main(); } catch (e) { console.error(String(e && e.message || e)); process.exit(1); }
