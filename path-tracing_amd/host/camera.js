'use strict';
/*
 * Camera -- host-side counterpart of /root/reference/classes/camera.js:3-92 without three.js.
 *
 * Same public surface: new Camera(fov, aspect, near, far), translate(axis, amount),
 * rotate(axis, angle), lookAt(x, y, z), getCamera() -> { cameraSource, cameraDirection, cameraUp,
 * cameraRight, cameraLeftBottom } (each a {x,y,z} record like the THREE.Vector3 fields main.js:292-296
 * reads).  fov/aspect/near/far are accepted and ignored exactly as in the reference: the image plane
 * is the unit square at distance 1 spanned by `right` and the WORLD up (0,1,0) (camera.js:65-81).
 *
 * three r163 behaviour restated here (parity unpinned, SURVEY.md 8c-3): a camera starts at the origin
 * looking down -Z with up (0,1,0); rotateX/Y/Z post-multiply the orientation quaternion by an
 * axis-angle quaternion (Object3D.rotateOnAxis); lookAt builds Matrix4.lookAt(position, target, up)
 * and takes its quaternion; getWorldDirection returns normalize(-column2 of matrixWorld).
 */
const { v3, quat, mat4 } = require('./math3');

const AXES = { x: [1, 0, 0], y: [0, 1, 0], z: [0, 0, 1] };
const rec = (a) => ({ x: a[0], y: a[1], z: a[2] });

class Camera {
  constructor(fov, aspect, near, far) {
    this.fov = fov; this.aspect = aspect; this.near = near; this.far = far; // unused, as in the reference
    this.position = [0, 0, 0];
    this.quaternion = quat.identity();
    this.up = [0, 1, 0];
    this.update();
  }

  translate(axisParameter, amount) {
    const k = 'xyz'.indexOf(String(axisParameter).toLowerCase());
    if (k < 0 || String(axisParameter).length !== 1) throw new Error();
    this.position[k] += amount;
    this.update();
  }

  rotate(axisParameter, angle) {
    const axis = AXES[String(axisParameter).toLowerCase()];
    if (!axis) throw new Error();
    this.quaternion = quat.multiply(this.quaternion, quat.fromAxisAngle(axis, angle));
    this.update();
  }

  lookAt(x, y, z) {
    this.quaternion = quat.fromRotationMatrix(mat4.lookAt(this.position, [x, y, z], this.up));
    this.update();
  }

  update() {
    const world = mat4.compose(this.position, this.quaternion, [1, 1, 1]);
    const source = this.position.slice();
    const direction = v3.normalize([-world[8], -world[9], -world[10]]);
    const up = this.up.slice();
    const right = v3.normalize(v3.cross(direction, up));
    const middle = v3.add(source, direction);
    const leftBottom = v3.sub(v3.sub(middle, v3.scale(right, 0.5)), v3.scale(up, 0.5));
    this.camera = {
      cameraSource: rec(source), cameraDirection: rec(direction), cameraUp: rec(up),
      cameraRight: rec(right), cameraLeft: rec(v3.scale(right, -1)), cameraMiddle: rec(middle),
      cameraLeftBottom: rec(leftBottom),
    };
    return this.camera;
  }

  getCamera() { return this.camera; }
}

/* Camera presets of main.js:146-171 plus the C3 preset SURVEY.md 8d asks for. */
const PRESETS = {
  'room-v3': (c) => { c.translate('x', 14); c.translate('z', -14); c.translate('y', 3); c.lookAt(0, 0, 0); },
  'cornell': (c) => { c.translate('x', 12.4); c.rotate('y', 3.14159265359 / 2); },
  'pixar-room': (c) => { c.translate('x', 2); c.translate('z', 4); c.translate('y', 2); c.lookAt(0, -2, -4); },
  'teapot': (c) => { c.translate('x', 1); c.translate('y', -1); c.lookAt(2, -10, -12); },
  // resources/cornell (2-unit box, opening towards +z): the committed presets do not frame it; not in the reference
  'cornell-box': (c) => { c.translate('y', 1); c.translate('z', 4.3); },
  // stand-alone teapot.obj (spans ~154 units): not in the reference, documented in DESIGN.md
  'teapot-obj': (c) => { c.translate('y', 40); c.translate('z', 260); c.lookAt(0, 0, 0); },
};

function cameraFromPreset(name, width, height) {
  const c = new Camera(50, width / height, 0.1, 1000);
  if (!PRESETS[name]) throw new Error('unknown camera preset ' + name);
  PRESETS[name](c);
  return c;
}

module.exports = { Camera, cameraFromPreset, PRESETS };
