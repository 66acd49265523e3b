#!/usr/bin/env python3
"""An independent ISO 11172-3 / 13818-3 Layer-III decoder for fixtures: FFmpeg's mpegaudiodec, as embedded in the
Chromium build that ships inside the `kaleido` wheel of this image.  BUILD CONTAINER ONLY: nothing here travels to the
GPU box and no product or test path imports this file; its OUTPUTS are committed as fixtures under tests/golden/
(tools/make_iso_golden.py writes them).

How: `kaleido` is a headless Chromium that reads one JSON request per line on stdin, calls
`kaleido_scopes.<scope>(request)` in a page and prints the promise's result as one JSON line.  The scope script is
read from ./js/kaleido_scopes.js relative to the working directory, so a temporary directory with symbolic links to
the wheel's bin/ lib/ etc/ xdg/ and OUR js/kaleido_scopes.js turns it into a decodeAudioData server:
  request  {"data": {"mp3": <base64>, "rate": 44100, "channels": 2}}
  response {"code": 0, "rate": .., "length": .., "channels": [<base64 float32 LE>, ...]}
The OfflineAudioContext is created AT THE FILE'S OWN RATE, so WebAudio does not resample: the floats are FFmpeg's
output (float planar, `mpegaudiodec_float`), scaled so that full scale is 1.0.

Start-of-stream handling (measured, see `align()`): Chromium's FFmpeg demuxer drops a leading Xing/Info frame and the
decoder's output starts at the stream's first sample, WITHOUT the 528+1-sample decoder delay trimmed when there is no
LAME tag; `align()` searches the offset instead of assuming it, and callers state what they found.
"""
import base64
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np

KALEIDO_DIR = "/usr/local/lib/python3.10/dist-packages/kaleido/executable"

SCOPE_JS = r"""
(function(){
  function b64ToBuf(s) {
    var bin = atob(s), n = bin.length, u = new Uint8Array(n);
    for (var i = 0; i < n; i++) u[i] = bin.charCodeAt(i);
    return u.buffer;
  }
  function bufToB64(f32) {
    var u = new Uint8Array(f32.buffer, f32.byteOffset, f32.byteLength), parts = [], CH = 0x8000;
    for (var i = 0; i < u.length; i += CH) parts.push(String.fromCharCode.apply(null, u.subarray(i, i + CH)));
    return btoa(parts.join(""));
  }
  function decode(info) {
    var d = info.data;
    return new Promise(function(resolve) {
      try {
        var ctx = new OfflineAudioContext(d.channels || 2, 1, d.rate);
        ctx.decodeAudioData(b64ToBuf(d.mp3), function(ab) {
          var ch = [];
          for (var c = 0; c < ab.numberOfChannels; c++) ch.push(bufToB64(ab.getChannelData(c)));
          resolve({code: 0, rate: ab.sampleRate, length: ab.length, channels: ch});
        }, function(e) { resolve({code: 1, message: "decodeAudioData: " + e}); });
      } catch (e) { resolve({code: 2, message: "" + e}); }
    });
  }
  window.kaleido_scopes = {plotly: decode};
})();
"""


class FFmpegRef:
    """One kaleido process; `decode(mp3_bytes, rate, channels)` -> float32 array [channels, samples]."""

    def __init__(self):
        if not os.path.isdir(KALEIDO_DIR):
            raise RuntimeError("kaleido is not in this image: " + KALEIDO_DIR)
        self.dir = tempfile.mkdtemp(prefix="ffref_")
        for sub in ("bin", "lib", "etc", "xdg"):
            os.symlink(os.path.join(KALEIDO_DIR, sub), os.path.join(self.dir, sub))
        os.mkdir(os.path.join(self.dir, "js"))
        with open(os.path.join(self.dir, "js", "kaleido_scopes.js"), "w") as f:
            f.write(SCOPE_JS)
        stub = os.path.join(self.dir, "plotly_stub.js")
        with open(stub, "w") as f:
            f.write("window.Plotly = {version: '2.0.0'};\n")
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.path.join(self.dir, "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
        env["FONTCONFIG_PATH"] = os.path.join(self.dir, "etc", "fonts")
        env["XDG_DATA_HOME"] = os.path.join(self.dir, "xdg")
        env.pop("LD_PRELOAD", None)
        self.proc = subprocess.Popen(
            ["./bin/kaleido", "plotly", "--plotlyjs=" + stub, "--disable-gpu", "--no-sandbox", "--single-process",
             "--allow-file-access-from-files", "--disable-breakpad", "--disable-dev-shm-usage"],
            cwd=self.dir, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        hello = self.proc.stdout.readline()
        if not hello:
            raise RuntimeError("kaleido did not start")
        st = json.loads(hello)
        if st.get("code", 0) != 0:
            raise RuntimeError("kaleido: %r" % st)

    def decode(self, mp3, rate, channels=2):
        req = {"data": {"mp3": base64.b64encode(bytes(mp3)).decode(), "rate": int(rate), "channels": int(channels)}}
        self.proc.stdin.write(json.dumps(req).encode() + b"\n")
        self.proc.stdin.flush()
        line = self.proc.stdout.readline()
        if not line:
            raise RuntimeError("kaleido died")
        r = json.loads(line)
        if r.get("code", 0) != 0:
            raise RuntimeError("decode failed: %r" % (r.get("message"),))
        if r["rate"] != rate:
            raise RuntimeError("resampled: asked %d, got %r" % (rate, r["rate"]))
        ch = [np.frombuffer(base64.b64decode(c), dtype="<f4") for c in r["channels"]]
        return np.stack(ch)

    def close(self):
        if self.proc is not None:
            try:
                self.proc.stdin.close()
                self.proc.wait(timeout=5)
            except Exception:
                self.proc.kill()
            self.proc = None
        shutil.rmtree(self.dir, ignore_errors=True)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def align(ours, theirs, max_shift=2400):
    """Offset d such that theirs[:, i] corresponds to ours[:, i + d] (both [channels, samples] float), by the
    smallest max-abs difference over the overlap; returns (d, maxabs).  Used once per fixture, and the offset is
    written into the fixture."""
    best = (None, np.inf)
    n = min(ours.shape[1], theirs.shape[1])
    probe = min(n - max_shift, 8192)
    for d in range(-max_shift, max_shift + 1):
        a0, b0 = (d, 0) if d >= 0 else (0, -d)
        a = ours[:, a0 + 1152:a0 + 1152 + probe]
        b = theirs[:, b0 + 1152:b0 + 1152 + probe]
        m = min(a.shape[1], b.shape[1])
        if m < 1152:
            continue
        e = np.abs(a[:, :m] - b[:, :m]).max()
        if e < best[1]:
            best = (d, e)
    return best


if __name__ == "__main__":
    import sys
    path = sys.argv[1]
    rate = int(sys.argv[2]) if len(sys.argv) > 2 else 44100
    with FFmpegRef() as ff:
        pcm = ff.decode(open(path, "rb").read(), rate)
    print(pcm.shape, float(np.abs(pcm).max()))
