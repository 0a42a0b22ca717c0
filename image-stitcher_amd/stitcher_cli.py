#!/usr/bin/env python3
"""Command line of the stitcher: the reference's flags (stitcher_cli.py:14-62) unchanged,
plus six switches for what this build adds (``--fusion-mode``, ``--normalization``,
``--zarr-compression``, ``--per-region-registration``, ``--flatfield-estimator``, ``--all-pairs-registration``).

    python -m image_stitcher_amd.stitcher_cli -i /path/to/acquisition -r -ff --registration-channel "488"
"""
import argparse
import sys

from .stitcher import Stitcher
from .stitcher_parameters import StitchingParameters


FLAGS = (
    # (names, kwargs) -- names and semantics as in the reference's argparse set-up (stitcher_cli.py:14-62)
    (('--input-folder', '-i'), dict(required=True, help="acquisition folder (timepoint sub-folders with tiles and coordinates.csv)")),
    (('--output-format', '-f'), dict(choices=['.ome.zarr', '.ome.tiff'], default='.ome.zarr', help="container of the stitched output")),
    (('--apply-flatfield', '-ff'), dict(action='store_true', help="divide every tile by its channel's flatfield")),
    (('--use-registration', '-r'), dict(action='store_true', help="register the centre tile pairs and place tiles by the measured shifts")),
    (('--registration-channel',), dict(help="channel the shifts are measured on (first channel when omitted)")),
    (('--registration-z-level',), dict(type=int, default=0, help="z plane the shifts are measured on")),
    (('--dynamic-registration',), dict(action='store_true', help="accepted, stored and ignored, exactly like the reference (stitcher.py:92); see --all-pairs-registration")),
    (('--scan-pattern', '-s'), dict(choices=['Unidirectional', 'S-Pattern'], default='Unidirectional', help="stage scan order")),
    (('--merge-timepoints', '-mt'), dict(action='store_true', help="request one dataset over all timepoints")),
    (('--merge-hcs-regions', '-mw'), dict(action='store_true', help="request one plate dataset over all wells")),
    (('--params-json',), dict(help="JSON file of StitchingParameters; replaces the flags above")),
    # additions of this build
    (('--fusion-mode',), dict(choices=['overwrite', 'feather'], default='overwrite',
                              help="overwrite = the reference's last-writer-wins; feather = distance-weighted blend (extension)")),
    (('--normalization',), dict(choices=['phase', 'none'], default='phase',
                                help="cross-power normalisation: phase = scikit-image >= 0.19 default, none = 0.18 behaviour")),
    (('--zarr-compression',), dict(choices=['blosc', 'zlib', 'none'], default='blosc',
                                   help="OME-Zarr chunk codec: blosc = the reference's default (Blosc-1 frames, shuffle + LZ4), encoded "
                                        "on the device; zlib = host threads; none = raw chunks")),
    (('--per-region-registration',), dict(action='store_true',
                                          help="with -r: register every (timepoint, region) on its own tiles instead of once")),
    (('--all-pairs-registration',), dict(action='store_true',
                                         help="with -r: register EVERY adjacent tile pair of the registration plane (batched on the device, "
                                              "sharded by pair over the ranks) and place tiles by the per-axis median shift, instead of the "
                                              "reference's centre-tile pairs")),
    (('--flatfield-estimator',), dict(choices=['auto', 'basic', 'basicpy', 'mean'], default='auto',
                                      help="with -ff: basicpy's BaSiC fit when that package is installed (auto / basicpy), this "
                                           "build's device restatement of the published BaSiC fit (basic; what auto falls back "
                                           "to), or a plain smoothed mean (mean: not BaSiC)")),
)


def parse_args(argv=None) -> argparse.Namespace:
    parser = argparse.ArgumentParser(description="Squid tile stitcher, MI355X core")
    for names, kwargs in FLAGS:
        parser.add_argument(*names, **kwargs)
    return parser.parse_args(argv)


def create_params(args: argparse.Namespace) -> StitchingParameters:
    if args.params_json:
        return StitchingParameters.from_json(args.params_json)
    return StitchingParameters.from_dict({
        'input_folder': args.input_folder, 'output_format': args.output_format,
        'apply_flatfield': args.apply_flatfield, 'use_registration': args.use_registration,
        'registration_channel': args.registration_channel, 'registration_z_level': args.registration_z_level,
        'scan_pattern': args.scan_pattern, 'merge_timepoints': args.merge_timepoints,
        'merge_hcs_regions': args.merge_hcs_regions, 'dynamic_registration': args.dynamic_registration})


def init_distributed():
    """Under torchrun (WORLD_SIZE > 1): one process per GPU, RCCL process group."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    local = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('SQ_DIST_BACKEND', 'nccl')
    device = torch.device('cuda', local % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if not dist.is_initialized():
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
    return device


def main(argv=None):
    args = parse_args(argv)
    try:
        device = init_distributed()
        params = create_params(args)
        stitcher = Stitcher(params, device=device, fusion_mode=args.fusion_mode,
                            normalization=None if args.normalization == 'none' else 'phase',
                            zarr_compression=args.zarr_compression,
                            per_region_registration=args.per_region_registration,
                            flatfield_estimator=args.flatfield_estimator,
                            all_pairs_registration=args.all_pairs_registration)
        print("Starting stitching with parameters:")
        for k, v in params.to_dict().items():
            print(f"{k}: {v}")
        stitcher.run()
    except Exception as e:   # same contract as the reference: message on stderr, exit 1
        print(f"Error: {e}", file=sys.stderr)
        sys.exit(1)


if __name__ == '__main__':
    main()
