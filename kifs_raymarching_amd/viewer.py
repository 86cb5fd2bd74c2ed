"""Optional live-viewer hook: input events in, finished frames out.

The reference's window loop has two halves around the render path.  Input: a pressed left button makes
the camera rotatable, mouse motion then rotates it by (-dx / 10, dy / 10) degrees, the wheel zooms by
its line delta (pixel deltas / 10), and each of them requests a redraw (render.rs:222-270).  Output:
the redraw renders and presents the surface (render.rs:354-356).  `ViewerSession` is that loop without
a window: the same handlers drive any object with GraphicState's camera methods, `redraw()` renders
when something changed and hands the RGBA8 frame to its sinks.

Sinks are what "present" means for a headless renderer:
  PngSequenceSink  numbered PNG files in a directory;
  HttpSink         a tiny HTTP server (standard library only, loopback by default): `/` is a page that
                   shows `/stream` (multipart PNG, one part per presented frame) and sends the
                   browser's mouse events to `/input`, which feeds the session's handlers; `/frame.png`
                   is the latest frame.
tools/view.py wires a GraphicState, a session and an HttpSink together.
"""
import threading
import time
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from pathlib import Path
from urllib.parse import parse_qs, urlparse

import numpy as np

from .image import png_bytes


class PngSequenceSink:
    """present(frame, index) -> <directory>/<prefix><index:05d>.png"""

    def __init__(self, directory, prefix: str = "frame_"):
        self.directory = Path(directory)
        self.directory.mkdir(parents=True, exist_ok=True)
        self.prefix = prefix
        self.written = []

    def present(self, frame: np.ndarray, index: int):
        path = self.directory / f"{self.prefix}{index:05d}.png"
        path.write_bytes(png_bytes(frame))
        self.written.append(path)

    def close(self):
        pass


_PAGE = b"""<!doctype html><title>kifs viewer</title>
<body style="margin:0;background:#111"><img id="v" src="/stream" draggable="false" style="max-width:100vw;max-height:100vh">
<script>
const v = document.getElementById('v'), send = q => fetch('/input?' + q, {method: 'POST'});
v.addEventListener('mousedown', e => { if (e.button === 0) send('button=1'); });
window.addEventListener('mouseup', e => { if (e.button === 0) send('button=0'); });
window.addEventListener('mousemove', e => { if (e.buttons & 1) send('dx=' + e.movementX + '&dy=' + e.movementY); });
window.addEventListener('wheel', e => send('pixels=' + (-e.deltaY)));
</script></body>"""


class HttpSink:
    """Serves presented frames over HTTP and routes `/input` requests to `on_input(dict)`.

    `/input` parameters: button=0|1 (left button released / pressed), dx=, dy= (mouse motion in
    pixels), lines= (wheel, line delta) or pixels= (wheel, pixel delta).  port=0 picks a free port
    (see `.port`)."""

    def __init__(self, host: str = "127.0.0.1", port: int = 8080, on_input=None, png_level: int = 1):
        self.on_input = on_input
        self.png_level = png_level
        self._latest = None      # (index, png bytes)
        self._cond = threading.Condition()
        self._closed = False
        sink = self

        class Handler(BaseHTTPRequestHandler):
            protocol_version = "HTTP/1.1"

            def log_message(self, *args):  # quiet
                pass

            def _send(self, code, ctype=None, body=b""):
                self.send_response(code)
                if ctype:
                    self.send_header("Content-Type", ctype)
                self.send_header("Content-Length", str(len(body)))
                self.send_header("Cache-Control", "no-store")
                self.end_headers()
                self.wfile.write(body)

            def do_GET(self):
                url = urlparse(self.path)
                if url.path == "/":
                    self._send(200, "text/html", _PAGE)
                elif url.path == "/frame.png":
                    latest = sink._latest
                    self._send(200, "image/png", latest[1]) if latest else self._send(404)
                elif url.path == "/stream":
                    self.send_response(200)
                    self.send_header("Content-Type", "multipart/x-mixed-replace; boundary=kifsframe")
                    self.send_header("Cache-Control", "no-store")
                    self.send_header("Connection", "close")
                    self.end_headers()
                    seen = -1
                    try:
                        while True:
                            with sink._cond:
                                sink._cond.wait_for(lambda: sink._closed or (sink._latest and sink._latest[0] != seen),
                                                    timeout=1.0)
                                if sink._closed:
                                    return
                                latest = sink._latest
                            if latest is None or latest[0] == seen:
                                continue
                            seen = latest[0]
                            self.wfile.write(b"--kifsframe\r\nContent-Type: image/png\r\nContent-Length: %d\r\n\r\n"
                                             % len(latest[1]) + latest[1] + b"\r\n")
                            self.wfile.flush()
                    except (BrokenPipeError, ConnectionResetError):
                        return
                elif url.path == "/input":
                    self._input(url)
                else:
                    self._send(404)

            def do_POST(self):
                url = urlparse(self.path)
                self._input(url) if url.path == "/input" else self._send(404)

            def _input(self, url):
                try:
                    event = {k: float(v[-1]) for k, v in parse_qs(url.query).items()}
                except ValueError:
                    return self._send(400)
                if sink.on_input:
                    sink.on_input(event)
                self._send(204)

        self._server = ThreadingHTTPServer((host, port), Handler)
        self._server.daemon_threads = True
        self.host, self.port = self._server.server_address[:2]
        self._thread = threading.Thread(target=self._server.serve_forever, name="kifs-viewer-http", daemon=True)
        self._thread.start()

    @property
    def url(self) -> str:
        return f"http://{self.host}:{self.port}/"

    def present(self, frame: np.ndarray, index: int):
        data = png_bytes(frame, level=self.png_level)
        with self._cond:
            self._latest = (index, data)
            self._cond.notify_all()

    def close(self):
        with self._cond:
            self._closed = True
            self._cond.notify_all()
        self._server.shutdown()
        self._server.server_close()


class ViewerSession:
    """The reference's event handlers and redraw, around `renderer` (a GraphicState, or anything with
    enable_camera_rotation / disable_camera_rotation / mouse_motion(dx, dy) / zoom_camera(distance) /
    render() -> (H, W, 4) uint8)."""

    def __init__(self, renderer, sinks=()):
        self.renderer = renderer
        self.sinks = list(sinks)
        self.frames_presented = 0
        self.last_frame_ms = 0.0
        self._dirty = True  # the first redraw always draws
        self._lock = threading.Lock()

    # ---- input (render.rs:222-270)
    def mouse_button(self, pressed: bool):
        with self._lock:
            if pressed:
                self.renderer.enable_camera_rotation()
            else:
                self.renderer.disable_camera_rotation()

    def mouse_wheel(self, lines: float = None, pixels: float = None):
        distance = float(lines) if lines is not None else float(pixels) / 10.0
        with self._lock:
            self.renderer.zoom_camera(distance)
            self._dirty = True

    def mouse_motion(self, dx: float, dy: float):
        with self._lock:
            if self.renderer.is_camera_rotatable():
                self.renderer.mouse_motion(dx, dy)
                self._dirty = True

    def request_redraw(self):
        with self._lock:
            self._dirty = True

    def handle(self, event: dict):
        """One `/input` request of HttpSink."""
        if "button" in event:
            self.mouse_button(event["button"] != 0)
        if "dx" in event or "dy" in event:
            self.mouse_motion(event.get("dx", 0.0), event.get("dy", 0.0))
        if "lines" in event:
            self.mouse_wheel(lines=event["lines"])
        elif "pixels" in event:
            self.mouse_wheel(pixels=event["pixels"])

    # ---- output (render.rs:300-360)
    def redraw(self) -> bool:
        """Renders and presents if anything changed since the last redraw."""
        with self._lock:
            if not self._dirty:
                return False
            self._dirty = False
            t0 = time.perf_counter()
            frame = self.renderer.render()
            self.last_frame_ms = (time.perf_counter() - t0) * 1e3
            index = self.frames_presented
            self.frames_presented += 1
        for s in self.sinks:
            s.present(frame, index)
        return True

    def run(self, seconds: float = None, idle_sleep: float = 0.005):
        """Redraw loop: until `seconds` have passed (None: until interrupted)."""
        end = None if seconds is None else time.monotonic() + seconds
        try:
            while end is None or time.monotonic() < end:
                if not self.redraw():
                    time.sleep(idle_sleep)
        except KeyboardInterrupt:
            pass

    def close(self):
        for s in self.sinks:
            s.close()
