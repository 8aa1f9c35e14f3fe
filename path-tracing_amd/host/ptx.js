'use strict';
/*
 * PathTracer -- JS face of the C-ABI addon (csrc/ptx_napi.node over libptx_hip.so, include/ptx.h).
 * It stands where the reference's main.js drives WebGL: uploadScene ~ the eleven uploadTexture() calls
 * (main.js:272-282), setCamera/setParams ~ the uniform block (main.js:290-308), renderFrames ~ the tile loop of
 * gl.drawArrays (main.js:330-355), read* ~ gl.readPixels (main.js:459).  No WebGL or CPU fallback: if the addon or
 * a gfx950 device is missing, construction throws.
 */
const path = require('path');

function loadAddon() {
  const file = path.join(__dirname, '..', 'csrc', 'ptx_napi.node');
  try { return require(file); } catch (e) {
    throw new Error('ptx: cannot load ' + file + ' (build it with `make -C path-tracing_amd/csrc`): ' + e.message);
  }
}

class PathTracer {
  /** device: one HIP device ordinal, or an array of them: one PathTracer over several GPUs (frames are dealt round-robin
   *  to the devices and the per-device float buffers are summed by one RCCL all-reduce when the image is read;
   *  include/ptx.h ptx_create). */
  constructor(device) {
    this.addon = loadAddon();
    this.handle = this.addon.create(Array.isArray(device) ? device.map((d) => d | 0) : (device || 0));
    this.width = 0; this.height = 0;
    this._busy = false;
    // one context, one thread: while an async render runs on the libuv pool, calls from the event loop are refused
    for (const name of Object.getOwnPropertyNames(PathTracer.prototype)) {
      if (name === 'constructor' || name === 'renderFramesAsync') continue;
      const fn = this[name];
      if (typeof fn !== 'function') continue;
      this[name] = (...args) => {
        if (this._busy) throw new Error('ptx: ' + name + '() while renderFramesAsync is running; await its promise first');
        return fn.apply(this, args);
      };
    }
  }

  /** scene: the typed arrays of host/scene.js buildSceneArrays() (or scenefile.readScene()). */
  uploadScene(scene) { this.addon.uploadScene(this.handle, scene); }

  /** camera: Camera.getCamera() record (classes/camera.js shape); values are rounded to f32 at the boundary. */
  setParams(width, height, camera, maxPathLength, sampleCount) {
    const v = (p) => (Array.isArray(p) ? p : [p.x, p.y, p.z]);
    this.addon.setParams(this.handle, {
      width, height, camSrc: v(camera.cameraSource), camUp: v(camera.cameraUp), camRight: v(camera.cameraRight),
      camLeftBottom: v(camera.cameraLeftBottom), maxPathLength, sampleCount });
    this.width = width; this.height = height;
  }

  setOption(key, value) { this.addon.setOption(this.handle, key, value); }
  /** timestamps: one int per frame (uniform `timestamp`, main.js:300). Blocks until the frames are done. */
  renderFrames(timestamps, firstFrameNumber) { this.addon.renderFrames(this.handle, Int32Array.from(timestamps), firstFrameNumber || 0); }
  /** Same, without blocking the event loop (the reference's frame loop is async, main.js:391): resolves when the frames are
   *  done.  The context is not thread-safe: until the promise has settled every other method of this PathTracer throws. */
  renderFramesAsync(timestamps, firstFrameNumber) {
    if (this._busy) return Promise.reject(new Error('ptx: a renderFramesAsync call is still running on this PathTracer'));
    this._busy = true;
    const done = () => { this._busy = false; };
    let pr;
    try { pr = this.addon.renderFramesAsync(this.handle, Int32Array.from(timestamps), firstFrameNumber || 0); } catch (e) { done(); throw e; }
    return pr.then((v) => { done(); return v; }, (e) => { done(); throw e; });
  }
  resetAccum() { this.addon.resetAccum(this.handle); }
  get framesDone() { return this.addon.framesDone(this.handle); }
  /** devices (ranks) whose buffers the all-reduce sums: the device count of a multi-GPU PathTracer, else 0 */
  get commWorld() { return this.addon.commWorld(this.handle); }
  // read-back buffers are sized by the addon from the context's own width and height
  readLinear() { return this.addon.readLinear(this.handle); }
  readSum() { return this.addon.readSum(this.handle); }
  readRgba8(gamma, flipY) { return this.addon.readRgba8(this.handle, gamma !== false, flipY !== false); }
  /** The reference's RGBA8 ping-pong target (setOption('compat_rgba8', 1 | 2) before rendering): what gl.readPixels returns in the browser
   *  (1 = GL's preferred store / NEAREST rules, 2 = SwiftShader's; include/ptx.h). */
  readCompatRgba8(gamma, flipY) { return this.addon.readCompatRgba8(this.handle, gamma === true, flipY === true); }
  /** Checkpoint restore: accumulation := sum (Float32Array W*H*3 from readSum()), holding framesDone frames. */
  writeSum(sum, framesDone) { this.addon.writeSum(this.handle, sum, framesDone); }
  traceClosest(origins, directions) { return this.addon.traceClosest(this.handle, Float32Array.from(origins), Float32Array.from(directions)); }
  primaryHits() { return this.addon.primaryHits(this.handle); }
  counters() { return this.addon.counters(this.handle); }
  resetCounters() { this.addon.resetCounters(this.handle); }
  destroy() { if (this.handle) { this.addon.destroy(this.handle); this.handle = null; } }
}

module.exports = { PathTracer, loadAddon };
