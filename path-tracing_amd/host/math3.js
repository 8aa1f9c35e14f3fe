'use strict';
/*
 * Minimal f64 vector / quaternion / 4x4 math for the JS host.
 *
 * The reference host leans on three@0.163.0 (package-lock.json:774-778), which is not vendored and
 * not installable offline.  These helpers restate the documented r163 semantics of exactly the
 * calls the hot path's callers make (SURVEY.md 8c-3): Vector3 {applyMatrix4, cross, normalize},
 * Matrix4 {compose, decompose, multiplyMatrices, lookAt}, Quaternion {setFromAxisAngle, multiply,
 * setFromRotationMatrix}.  Column-major element order (e[col*4+row]) like three.  No reference test
 * pins any of them: parity unpinned at this boundary.
 */

const v3 = {
  make: (x, y, z) => [x, y, z],
  add: (a, b) => [a[0] + b[0], a[1] + b[1], a[2] + b[2]],
  sub: (a, b) => [a[0] - b[0], a[1] - b[1], a[2] - b[2]],
  scale: (a, s) => [a[0] * s, a[1] * s, a[2] * s],
  dot: (a, b) => a[0] * b[0] + a[1] * b[1] + a[2] * b[2],
  cross: (a, b) => [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]],
  length: (a) => Math.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]),
  // Vector3.normalize(): divideScalar(length || 1) == multiplyScalar(1 / (length || 1))
  normalize: (a) => { const l = v3.length(a) || 1; return v3.scale(a, 1 / l); },
  min: (a, b) => [Math.min(a[0], b[0]), Math.min(a[1], b[1]), Math.min(a[2], b[2])],
  max: (a, b) => [Math.max(a[0], b[0]), Math.max(a[1], b[1]), Math.max(a[2], b[2])],
  // Vector3.applyMatrix4: affine transform with the perspective divide three performs
  applyMatrix4: (p, e) => {
    const x = p[0], y = p[1], z = p[2];
    const w = 1 / (e[3] * x + e[7] * y + e[11] * z + e[15]);
    return [(e[0] * x + e[4] * y + e[8] * z + e[12]) * w,
            (e[1] * x + e[5] * y + e[9] * z + e[13]) * w,
            (e[2] * x + e[6] * y + e[10] * z + e[14]) * w];
  },
};

const quat = {
  identity: () => [0, 0, 0, 1],
  fromAxisAngle: (axis, angle) => {
    const h = angle / 2, s = Math.sin(h);
    return [axis[0] * s, axis[1] * s, axis[2] * s, Math.cos(h)];
  },
  multiply: (a, b) => {
    const ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    return [ax * bw + aw * bx + ay * bz - az * by,
            ay * bw + aw * by + az * bx - ax * bz,
            az * bw + aw * bz + ax * by - ay * bx,
            aw * bw - ax * bx - ay * by - az * bz];
  },
  // Quaternion.setFromRotationMatrix (upper 3x3 of a column-major 4x4, assumed unscaled)
  fromRotationMatrix: (te) => {
    const m11 = te[0], m12 = te[4], m13 = te[8], m21 = te[1], m22 = te[5], m23 = te[9],
      m31 = te[2], m32 = te[6], m33 = te[10], trace = m11 + m22 + m33;
    if (trace > 0) {
      const s = 0.5 / Math.sqrt(trace + 1.0);
      return [(m32 - m23) * s, (m13 - m31) * s, (m21 - m12) * s, 0.25 / s];
    } else if (m11 > m22 && m11 > m33) {
      const s = 2.0 * Math.sqrt(1.0 + m11 - m22 - m33);
      return [0.25 * s, (m12 + m21) / s, (m13 + m31) / s, (m32 - m23) / s];
    } else if (m22 > m33) {
      const s = 2.0 * Math.sqrt(1.0 + m22 - m11 - m33);
      return [(m12 + m21) / s, 0.25 * s, (m23 + m32) / s, (m13 - m31) / s];
    }
    const s = 2.0 * Math.sqrt(1.0 + m33 - m11 - m22);
    return [(m13 + m31) / s, (m23 + m32) / s, 0.25 * s, (m21 - m12) / s];
  },
};

const mat4 = {
  identity: () => [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1],
  // Matrix4.compose(position, quaternion, scale)
  compose: (p, q, s) => {
    const x = q[0], y = q[1], z = q[2], w = q[3];
    const x2 = x + x, y2 = y + y, z2 = z + z;
    const xx = x * x2, xy = x * y2, xz = x * z2;
    const yy = y * y2, yz = y * z2, zz = z * z2;
    const wx = w * x2, wy = w * y2, wz = w * z2;
    const sx = s[0], sy = s[1], sz = s[2];
    return [(1 - (yy + zz)) * sx, (xy + wz) * sx, (xz - wy) * sx, 0,
            (xy - wz) * sy, (1 - (xx + zz)) * sy, (yz + wx) * sy, 0,
            (xz + wy) * sz, (yz - wx) * sz, (1 - (xx + yy)) * sz, 0,
            p[0], p[1], p[2], 1];
  },
  // Matrix4.multiplyMatrices(a, b) = a * b
  multiply: (a, b) => {
    const r = new Array(16);
    for (let c = 0; c < 4; c++) {
      for (let row = 0; row < 4; row++) {
        r[c * 4 + row] = a[row] * b[c * 4] + a[4 + row] * b[c * 4 + 1] + a[8 + row] * b[c * 4 + 2] +
          a[12 + row] * b[c * 4 + 3];
      }
    }
    return r;
  },
  determinant: (te) => {
    const n11 = te[0], n12 = te[4], n13 = te[8], n14 = te[12];
    const n21 = te[1], n22 = te[5], n23 = te[9], n24 = te[13];
    const n31 = te[2], n32 = te[6], n33 = te[10], n34 = te[14];
    const n41 = te[3], n42 = te[7], n43 = te[11], n44 = te[15];
    return (
      n41 * (+n14 * n23 * n32 - n13 * n24 * n32 - n14 * n22 * n33 + n12 * n24 * n33 + n13 * n22 * n34 - n12 * n23 * n34) +
      n42 * (+n11 * n23 * n34 - n11 * n24 * n33 + n14 * n21 * n33 - n13 * n21 * n34 + n13 * n24 * n31 - n14 * n23 * n31) +
      n43 * (+n11 * n24 * n32 - n11 * n22 * n34 - n14 * n21 * n32 + n12 * n21 * n34 + n14 * n22 * n31 - n12 * n24 * n31) +
      n44 * (-n13 * n22 * n31 - n11 * n23 * n32 + n11 * n22 * n33 + n13 * n21 * n32 - n12 * n21 * n33 + n12 * n23 * n31));
  },
  // Matrix4.decompose -> {position, quaternion, scale}
  decompose: (te) => {
    let sx = v3.length([te[0], te[1], te[2]]);
    const sy = v3.length([te[4], te[5], te[6]]);
    const sz = v3.length([te[8], te[9], te[10]]);
    if (mat4.determinant(te) < 0) sx = -sx;
    const m = te.slice();
    const ix = 1 / sx, iy = 1 / sy, iz = 1 / sz;
    m[0] *= ix; m[1] *= ix; m[2] *= ix;
    m[4] *= iy; m[5] *= iy; m[6] *= iy;
    m[8] *= iz; m[9] *= iz; m[10] *= iz;
    return { position: [te[12], te[13], te[14]], quaternion: quat.fromRotationMatrix(m), scale: [sx, sy, sz] };
  },
  // Matrix4.lookAt(eye, target, up): rotation whose +z points from target to eye
  lookAt: (eye, target, up) => {
    let z = v3.sub(eye, target);
    if (v3.dot(z, z) === 0) z[2] = 1;
    z = v3.normalize(z);
    let x = v3.cross(up, z);
    if (v3.dot(x, x) === 0) {
      if (Math.abs(up[2]) === 1) z[0] += 0.0001; else z[2] += 0.0001;
      z = v3.normalize(z);
      x = v3.cross(up, z);
    }
    x = v3.normalize(x);
    const y = v3.cross(z, x);
    return [x[0], x[1], x[2], 0, y[0], y[1], y[2], 0, z[0], z[1], z[2], 0, 0, 0, 0, 1];
  },
};

module.exports = { v3, quat, mat4 };
