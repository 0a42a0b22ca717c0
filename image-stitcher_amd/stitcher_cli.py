#!/usr/bin/env python3
"""Command line of the stitcher: the reference's flags (stitcher_cli.py:14-62) unchanged,
plus two switches for what this build adds (``--fusion-mode``, ``--normalization``).

    python -m image_stitcher_amd.stitcher_cli -i /path/to/acquisition -r -ff --registration-channel "488"
"""
import argparse
import sys

from .stitcher import Stitcher
from .stitcher_parameters import StitchingParameters


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Microscopy Image Stitching CLI (MI355X core)")
    p.add_argument('--input-folder', '-i', required=True, help="Input folder containing images to stitch")
    p.add_argument('--output-format', '-f', choices=['.ome.zarr', '.ome.tiff'], default='.ome.zarr',
                   help="Output format for stitched data (default: .ome.zarr)")
    p.add_argument('--apply-flatfield', '-ff', action='store_true', help="Apply flatfield correction")
    p.add_argument('--use-registration', '-r', action='store_true', help="Enable image registration")
    p.add_argument('--registration-channel', help="Channel to use for registration (default: first available channel)")
    p.add_argument('--registration-z-level', type=int, default=0, help="Z-level to use for registration (default: 0)")
    p.add_argument('--dynamic-registration', action='store_true', help="Use dynamic registration for improved accuracy")
    p.add_argument('--scan-pattern', '-s', choices=['Unidirectional', 'S-Pattern'], default='Unidirectional',
                   help="Microscope scanning pattern (default: Unidirectional)")
    p.add_argument('--merge-timepoints', '-mt', action='store_true', help="Merge all timepoints into a single dataset")
    p.add_argument('--merge-hcs-regions', '-mw', action='store_true',
                   help="Merge all high-content screening regions (wells)")
    p.add_argument('--params-json', help="Path to a JSON file containing stitching parameters (overrides other arguments)")
    # additions of this build
    p.add_argument('--fusion-mode', choices=['overwrite', 'feather'], default='overwrite',
                   help="overwrite = the reference's last-writer-wins; feather = distance-weighted blend (extension)")
    p.add_argument('--normalization', choices=['phase', 'none'], default='phase',
                   help="cross-power normalisation: phase = scikit-image >= 0.19 default, none = 0.18 behaviour")
    return p.parse_args(argv)


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
                            normalization=None if args.normalization == 'none' else 'phase')
        print("Starting stitching with parameters:")
        for k, v in params.to_dict().items():
            print(f"{k}: {v}")
        stitcher.run()
    except Exception as e:   # same contract as the reference: message on stderr, exit 1
        print(f"Error: {e}", file=sys.stderr)
        sys.exit(1)


if __name__ == '__main__':
    main()
