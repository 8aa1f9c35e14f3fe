/*
 * webgl_harness.js -- TEST INFRASTRUCTURE ONLY (used by oracle/gen_ref_frames.py in the build container).
 *
 * Drives the reference's shaders, VERBATIM (their text arrives in the request payload, read from
 * /root/reference/shaders at run time and never stored in this repo), on the SwiftShader WebGL2 of the headless
 * Chromium that ships with the `kaleido` Python package.  kaleido's "plotly" scope loads this file as if it
 * were plotly.js and calls window.Plotly.toImage(figure); the payload rides in figure.data[0].customdata[0]
 * and the returned string comes back on stdout (SURVEY.md Appendix A-1).
 *
 * What it reproduces from the reference host: RGB32F data textures exactly as utils/textureUploader.js:4-51
 * builds them (width min(4096,n), zero padded, NEAREST, CLAMP_TO_EDGE; integers triplicated into r,g,b), the
 * uniform block of main.js:290-308, an RGBA8 render target like modules/bufferManager.js:18, ONE full-viewport
 * gl.drawArrays(TRIANGLE_STRIP, 0, 4) (main.js:350 without the 32x32 tiling, which does not change results) and
 * gl.readPixels(RGBA, UNSIGNED_BYTE) (main.js:459).
 */
(function () {
  function compile(gl, type, src) {
    var s = gl.createShader(type);
    gl.shaderSource(s, src);
    gl.compileShader(s);
    if (!gl.getShaderParameter(s, gl.COMPILE_STATUS)) throw new Error('compile: ' + gl.getShaderInfoLog(s));
    return s;
  }
  function upload(gl, prog, data, name, unit) {
    var n = data.length / 3;
    var width = Math.min(4096, n), height = Math.ceil(n / 4096);
    var arr = new Float32Array(Math.max(width * height * 3, 3));
    arr.set(data);
    var tex = gl.createTexture();
    gl.activeTexture(gl.TEXTURE0 + unit);
    gl.bindTexture(gl.TEXTURE_2D, tex);
    gl.texImage2D(gl.TEXTURE_2D, 0, gl.RGB32F, Math.max(width, 1), Math.max(height, 1), 0, gl.RGB, gl.FLOAT, arr);
    gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_S, gl.CLAMP_TO_EDGE);
    gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_T, gl.CLAMP_TO_EDGE);
    gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MIN_FILTER, gl.NEAREST);
    gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MAG_FILTER, gl.NEAREST);
    gl.uniform1i(gl.getUniformLocation(prog, name), unit);
  }
  function trip(a) { var o = new Array(a.length * 3); for (var i = 0; i < a.length; i++) { o[3 * i] = a[i]; o[3 * i + 1] = a[i]; o[3 * i + 2] = a[i]; } return o; }
  function rgba8Target(gl, w, h, unit) {
    var tex = gl.createTexture();
    gl.activeTexture(gl.TEXTURE0 + unit);
    gl.bindTexture(gl.TEXTURE_2D, tex);
    gl.texImage2D(gl.TEXTURE_2D, 0, gl.RGBA, w, h, 0, gl.RGBA, gl.UNSIGNED_BYTE, null);
    gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MIN_FILTER, gl.NEAREST);
    gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MAG_FILTER, gl.NEAREST);
    return tex;
  }
  function b64(u8) { var s = ''; for (var i = 0; i < u8.length; i += 0x8000) s += String.fromCharCode.apply(null, u8.subarray(i, i + 0x8000)); return btoa(s); }

  function run(p) {
    var canvas = document.createElement('canvas');
    canvas.width = p.width; canvas.height = p.height;
    var gl = canvas.getContext('webgl2', { antialias: false, alpha: true, premultipliedAlpha: false });
    if (!gl) throw new Error('no webgl2');
    var info = { version: gl.getParameter(gl.VERSION), renderer: gl.getParameter(gl.RENDERER) };
    var dbg = gl.getExtension('WEBGL_debug_renderer_info');
    if (dbg) info.unmasked = gl.getParameter(dbg.UNMASKED_RENDERER_WEBGL);
    var prog = gl.createProgram();
    gl.attachShader(prog, compile(gl, gl.VERTEX_SHADER, p.vertexShader));
    gl.attachShader(prog, compile(gl, gl.FRAGMENT_SHADER, p.fragmentShader));
    gl.linkProgram(prog);
    if (!gl.getProgramParameter(prog, gl.LINK_STATUS)) throw new Error('link: ' + gl.getProgramInfoLog(prog));
    gl.useProgram(prog);
    gl.disable(gl.DITHER); gl.disable(gl.BLEND);

    var s = p.scene, unit = 2;
    upload(gl, prog, s.coords, 'coordinatesTexture', unit++);
    upload(gl, prog, s.normals, 'normalsTexture', unit++);
    upload(gl, prog, s.colors, 'colorsTexture', unit++);
    upload(gl, prog, s.emissions, 'emissionsTexture', unit++);
    upload(gl, prog, trip(s.light_tri), 'lightIndicesTexture', unit++);
    upload(gl, prog, s.node_min, 'nodesBoundingBoxesMins', unit++);
    upload(gl, prog, s.node_max, 'nodesBoundingBoxesMaxs', unit++);
    upload(gl, prog, trip(s.node_miss), 'nodesMissLinkIndices', unit++);
    upload(gl, prog, trip(s.node_count), 'nodesTrianglesCount', unit++);
    upload(gl, prog, trip(s.node_first), 'nodesInorderTrianglesIndices', unit++);
    upload(gl, prog, trip(s.inorder_tri), 'inorderTrianglesIndicesArray', unit++);

    var U = function (n) { return gl.getUniformLocation(prog, n); };
    gl.uniform2f(U('windowSize'), p.width, p.height);
    gl.uniform1f(U('aspectRatio'), p.width / p.height);
    gl.uniform3f(U('cameraSource'), p.cam_src[0], p.cam_src[1], p.cam_src[2]);
    gl.uniform3f(U('cameraDirection'), p.cam_dir[0], p.cam_dir[1], p.cam_dir[2]);
    gl.uniform3f(U('cameraUp'), p.cam_up[0], p.cam_up[1], p.cam_up[2]);
    gl.uniform3f(U('cameraRight'), p.cam_right[0], p.cam_right[1], p.cam_right[2]);
    gl.uniform3f(U('cameraLeftBottom'), p.cam_left_bottom[0], p.cam_left_bottom[1], p.cam_left_bottom[2]);
    gl.uniform1i(U('vertexCount'), s.T * 3);
    gl.uniform1i(U('triangleCount'), s.T);
    gl.uniform1i(U('lightIndicesCount'), s.L);
    gl.uniform1i(U('maxPathLength'), p.max_path_length);
    gl.uniform1i(U('sampleCount'), p.sample_count);
    gl.uniform1i(U('totalFrames'), p.timestamps.length);
    gl.uniform1i(U('bvhNodeCount'), s.N);
    gl.uniform1i(U('maxTextureSize'), 4096);
    gl.uniform1i(U('quadSize'), 32);
    gl.uniform1i(U('quadX'), 0);
    gl.uniform1i(U('quadY'), 0);

    // two RGBA8 targets with a framebuffer each, used in turn like modules/bufferManager.js:33-43 + main.js:312-324:
    // frame n renders into target n % 2 and samples target (n + 1) % 2 as previousFrameTexture
    var tex = [rgba8Target(gl, p.width, p.height, 0), rgba8Target(gl, p.width, p.height, 1)];
    var fbs = [];
    for (var k = 0; k < 2; k++) {
      var fbk = gl.createFramebuffer();
      gl.bindFramebuffer(gl.FRAMEBUFFER, fbk);
      gl.framebufferTexture2D(gl.FRAMEBUFFER, gl.COLOR_ATTACHMENT0, gl.TEXTURE_2D, tex[k], 0);
      if (gl.checkFramebufferStatus(gl.FRAMEBUFFER) !== gl.FRAMEBUFFER_COMPLETE) throw new Error('fbo incomplete');
      gl.clearColor(0, 0, 0, 1);
      gl.clear(gl.COLOR_BUFFER_BIT);
      fbs.push(fbk);
    }

    var vao = gl.createVertexArray();
    gl.bindVertexArray(vao);
    var vb = gl.createBuffer();
    gl.bindBuffer(gl.ARRAY_BUFFER, vb);
    gl.bufferData(gl.ARRAY_BUFFER, new Float32Array([-1, -1, 1, -1, -1, 1, 1, 1]), gl.STATIC_DRAW);
    var loc = gl.getAttribLocation(prog, 'position');
    gl.enableVertexAttribArray(loc);
    gl.vertexAttribPointer(loc, 2, gl.FLOAT, false, 0, 0);

    // p.outputFragmentShader: the reference's display pass (shaders/fragmentShaderOutput.glsl, verbatim) drawn into the canvas
    // after every frame exactly as main.js:357-373 does (default framebuffer, the frame's target bound as u_texture, full
    // viewport, clear, one TRIANGLE_STRIP), then gl.readPixels of the canvas like main.js:459 (before the row flip of :464-473)
    var progOut = null, vaoOut = null, display = [];
    if (p.outputFragmentShader) {
      progOut = gl.createProgram();
      gl.attachShader(progOut, compile(gl, gl.VERTEX_SHADER, p.vertexShader));
      gl.attachShader(progOut, compile(gl, gl.FRAGMENT_SHADER, p.outputFragmentShader));
      gl.linkProgram(progOut);
      if (!gl.getProgramParameter(progOut, gl.LINK_STATUS)) throw new Error('link(output): ' + gl.getProgramInfoLog(progOut));
      vaoOut = gl.createVertexArray();
      gl.bindVertexArray(vaoOut);
      gl.bindBuffer(gl.ARRAY_BUFFER, vb);
      var locOut = gl.getAttribLocation(progOut, 'position');
      gl.enableVertexAttribArray(locOut);
      gl.vertexAttribPointer(locOut, 2, gl.FLOAT, false, 0, 0);
      gl.bindVertexArray(vao);
    }

    var frames = [], times = [];
    // p.debug_words: a DERIVED shader (built by gen_ref_frames.py from the reference text at run time) writes one 32-bit word
    // per pixel as four bytes; one draw per word, selected by the uniform debugWord
    var nDraws = p.debug_words ? p.debug_words.length : p.timestamps.length;
    for (var f = 0; f < nDraws; f++) {
      if (p.debug_words) gl.uniform1i(U('debugWord'), p.debug_words[f]);
      // default: frameNumber = 0 for every frame: blend factor 1 => the un-blended per-frame image (SURVEY.md 8c-1
      // "Limits"); p.progressive: frameNumber = f, the reference's running blend through the 8-bit targets
      var n = p.progressive ? f : 0;
      gl.uniform1i(U('timestamp'), p.timestamps[p.debug_words ? 0 : f]);
      gl.uniform1i(U('frameNumber'), n);
      gl.activeTexture(gl.TEXTURE0); gl.bindTexture(gl.TEXTURE_2D, tex[(n + 1) % 2]);
      gl.uniform1i(U('previousFrameTexture'), 0);
      gl.bindFramebuffer(gl.FRAMEBUFFER, fbs[n % 2]);
      gl.viewport(0, 0, p.width, p.height);
      gl.clearColor(0, 0, 0, 1);
      gl.clear(gl.COLOR_BUFFER_BIT);
      var t0 = performance.now();
      gl.drawArrays(gl.TRIANGLE_STRIP, 0, 4);
      var px = new Uint8Array(p.width * p.height * 4);
      gl.readPixels(0, 0, p.width, p.height, gl.RGBA, gl.UNSIGNED_BYTE, px);
      times.push(performance.now() - t0);
      frames.push(b64(px));
      if (progOut) {
        gl.bindFramebuffer(gl.FRAMEBUFFER, null);
        gl.useProgram(progOut);
        gl.bindVertexArray(vaoOut);
        gl.activeTexture(gl.TEXTURE0 + (n % 2)); gl.bindTexture(gl.TEXTURE_2D, tex[n % 2]);
        gl.uniform1i(gl.getUniformLocation(progOut, 'u_texture'), n % 2);
        gl.viewport(0, 0, p.width, p.height);
        gl.clear(gl.COLOR_BUFFER_BIT);
        gl.drawArrays(gl.TRIANGLE_STRIP, 0, 4);
        var dpx = new Uint8Array(p.width * p.height * 4);
        gl.readPixels(0, 0, p.width, p.height, gl.RGBA, gl.UNSIGNED_BYTE, dpx);
        display.push(b64(dpx));
        gl.useProgram(prog);
        gl.bindVertexArray(vao);
      }
    }
    return { info: info, frames: frames, display: display, ms: times, glError: gl.getError() };
  }

  window.Plotly = {
    version: '2.0.0',
    toImage: function (fig, opts) {
      return new Promise(function (resolve) {
        try { resolve(JSON.stringify(run(fig.data[0].customdata[0]))); }
        catch (e) { resolve(JSON.stringify({ error: String(e && e.stack || e) })); }
      });
    }
  };
})();
