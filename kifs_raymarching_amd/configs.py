"""The BASELINE.json workloads as scene descriptions (synthetic, deterministic).

Common settings: camera distance 5, phi = theta = 0
(CameraData::default, data.rs:105-113), max_distance 1000, epsilon 1e-4, fractal colour
sRGB 200 via the reference's /256 conversion, black background, sRGB colour target.
`iters` = (sdf_iters, normal_iters, fold_iters); the reference constants are (100, 10, 10).
"""
import math
from dataclasses import dataclass, field

from .graphics import CameraData, FractalGroup, GuiData, PrimitiveShape, ScreenData

JULIA_C = (-0.2, 0.6, 0.2, 0.2)  # (real, i, j, k) of BASELINE configs 1, 2, 4


@dataclass
class Workload:
    name: str
    screen: ScreenData
    gui: GuiData
    iters: tuple
    camera: CameraData = field(default_factory=CameraData)
    frames: int = 1
    gpus: int = 1
    extensions: dict = None  # keyword arguments of GraphicState.set_extensions (soft shadows)

    @property
    def pixels(self):
        return self.screen.width * self.screen.height


def _julia(max_iterations, c=JULIA_C):
    return GuiData(max_iterations=max_iterations, fractal_group=FractalGroup.JuliaSet, constant=c)


def _sierpinski(max_iterations):
    return GuiData(max_iterations=max_iterations, fractal_group=FractalGroup.KaleidoscopicIFS,
                   primitive_shape=PrimitiveShape.SierpinskiTetrahedron)


WORKLOADS = {
    # configs[0]: the reference's own CPU-runnable plumbing case
    "cfg1_julia_256": Workload("256x256 quaternion-Julia, 64 steps, 8 SDF iters",
                               ScreenData(256, 256), _julia(64), (8, 10, 10)),
    # configs[1]: the headline metric
    "cfg2_julia_1080p": Workload("1920x1080 quaternion-Julia, 256 steps, 12 SDF iters",
                                 ScreenData(1920, 1080), _julia(256), (12, 10, 10)),
    "cfg3_sierpinski_1080p": Workload("1920x1080 KIFS Sierpinski, 16 folds, 256 steps",
                                      ScreenData(1920, 1080), _sierpinski(256), (100, 10, 16)),
    "cfg4_julia_4096": Workload("4096x4096 quaternion-Julia, 512 steps, 16 SDF iters",
                                ScreenData(4096, 4096), _julia(512), (16, 10, 10), gpus=8),
    # configs[4] without the soft-shadow extension (absent from the reference)
    "cfg5_sierpinski_8k_orbit": Workload("7680x4320 KIFS Sierpinski orbit, 16 folds, 256 steps",
                                         ScreenData(7680, 4320), _sierpinski(256), (100, 10, 16),
                                         camera=CameraData(origin_distance=3.0, theta=0.3),
                                         frames=120, gpus=8),
    # configs[4] with the soft-shadow extension switched on (no reference counterpart)
    "cfg5_sierpinski_8k_orbit_shadows": Workload(
        "7680x4320 KIFS Sierpinski orbit, 16 folds, 256 steps, soft-shadow secondary rays",
        ScreenData(7680, 4320), _sierpinski(256), (100, 10, 16),
        camera=CameraData(origin_distance=3.0, theta=0.3), frames=120, gpus=8,
        extensions=dict(soft_shadow=True, shadow_steps=64, shadow_k=8.0, shadow_t0=0.02, shadow_max_t=10.0)),
    # the reference exactly as shipped: GUI-default constant, hard-coded iteration counts
    "ref_julia_1080p": Workload("1920x1080 Julia, reference constants (100/10), GUI default c",
                                ScreenData(1920, 1080),
                                _julia(256, c=(-0.1, 0.6, 0.9, -0.3)), (100, 10, 10)),
    # the reference's third pipeline and its last primitive
    "n1_genjulia_1080p": Workload("1920x1080 generalised Julia, power 8, reference constants (100/10)",
                                  ScreenData(1920, 1080),
                                  GuiData(max_iterations=256, fractal_group=FractalGroup.GeneralizedJuliaSet,
                                          power=8.0, constant=(-0.1, 0.6, 0.9, -0.3)), (100, 10, 10)),
    "n2_bunny_1080p": Workload("1920x1080 KIFS bunny (neural SDF), 256 steps",
                               ScreenData(1920, 1080),
                               GuiData(max_iterations=256, fractal_group=FractalGroup.KaleidoscopicIFS,
                                       primitive_shape=PrimitiveShape.Bunny), (100, 10, 10)),
}

HEADLINE = "cfg2_julia_1080p"


def orbit_camera(workload: Workload, frame: int) -> CameraData:
    """Frame k of the cfg-5 orbit: phi_k = 2*pi*k/frames, fixed theta and distance."""
    base = workload.camera
    frames = max(workload.frames, 120)
    return CameraData(origin_distance=base.origin_distance, min_distance=base.min_distance,
                      phi=2.0 * math.pi * frame / frames, theta=base.theta)
