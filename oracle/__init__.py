"""CPU oracle for the HoloDiffusion denoise-and-render hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``holo_diffusion_amd`` (the product) may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

Pinning status
--------------
* Denoiser half (``unet_oracle``, ``diffusion_oracle``): PINNED.  The reference's
  own ``holo_diffusion/guided_diffusion/{unet,nn,gaussian_diffusion}.py`` import
  in the development container (torch + numpy only); ``oracle/make_golden.py``
  runs them on seeded inputs and commits the outputs under ``tests/golden/``.
  ``tests/test_oracle_golden.py`` checks the restatement against those vectors.
* Renderer half (``render_oracle``): PARITY UNPINNED.  Every reference file on
  that half imports PyTorch3D 0.7.4 (``environment.yaml:139``), which is an
  un-vendored third-party dependency absent from ``/root/reference`` and not
  installable here.  The restatement follows the reference call sites
  (``holo_voxel_grid_implicit_function.py:182-269``, ``custom_modules.py:44-160``,
  ``holo_multipass_ea.py:79-125``, ``configs/apple.yaml:135-165``) and PyTorch3D's
  published algorithms (NDC ray sampler, ``EmissionAbsorptionRaymarcher``,
  ``RayPointRefiner``/``sample_pdf``, ``VolumeLocator`` + ``F.grid_sample``);
  it is anchored by analytic known-answer tests and by ``torch.nn.functional``
  primitives that exist on both boxes.
"""
