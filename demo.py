"""demo.py -- same CLI as the reference's omnidata_tools/torch/demo.py:23-34:

    python demo.py --task {normal,depth} --img_path <file-or-dir> --output_path <dir>

It reads ./pretrained_models/omnidata_dpt_{normal,depth}_v2.ckpt (demo.py:36,62,80), writes
<stem>_<task>.png and <stem>_rgb.png (demo.py:127,134) and iterates glob(img_path+'/*') for a
directory (demo.py:158-160).  Extras for offline use: --weights PATH, --random-weights SEED,
--dtype (default 'mixed': within 1e-3 of the reference's fp32 forward; bf16 / fp16 / fp8 are faster throughput
modes that are not).  The forward runs on an MI355X through libdptx.so; no CPU fallback.
"""
import argparse
import glob
import os
import sys
from pathlib import Path

import torch
from PIL import Image


def main(argv=None):
    parser = argparse.ArgumentParser(description="Visualize output for depth or surface normals")
    parser.add_argument("--task", dest="task", help="normal or depth")
    parser.set_defaults(task="NONE")
    parser.add_argument("--img_path", dest="img_path", help="path to rgb image")
    parser.add_argument("--output_path", dest="output_path", help="path to where output image should be stored")
    parser.add_argument("--weights", default=None, help="checkpoint path (default ./pretrained_models/omnidata_dpt_<task>_v2.ckpt)")
    parser.add_argument("--random-weights", type=int, default=None, metavar="SEED", help="seeded synthetic weights (offline)")
    parser.add_argument("--dtype", default="mixed", choices=["mixed", "fp16x3", "bf16x3", "fp16", "bf16", "fp8"],
                        help="mixed (default) matches the reference within 1e-3; bf16 / fp16 / fp8 are ~2x faster and do not")
    parser.add_argument("--backbone", default="vitb_rn50_384", choices=["vitb_rn50_384", "vitl16_384"],
                        help="vitb_rn50_384 = DPT-Hybrid (the v2 checkpoints); vitl16_384 = DPT-Large (demo.py:81, the v1 depth model)")
    args = parser.parse_args(argv)

    if args.task not in ("normal", "depth"):
        print("task should be one of the following: normal, depth")
        sys.exit()
    if args.img_path is None or args.output_path is None:
        print("invalid file path!")
        sys.exit()

    from omnidata_amd.model import build_model
    from omnidata_amd import preprocess as pp

    os.makedirs(args.output_path, exist_ok=True)
    if not torch.cuda.is_available():
        raise RuntimeError("demo.py needs an AMD GPU: the DPT forward is implemented as HIP kernels only")
    device = torch.device("cuda:0")
    weights = args.weights
    if weights is None and args.random_weights is None:
        weights = "./pretrained_models/" + ("omnidata_dpt_normal_v2.ckpt" if args.task == "normal" else "omnidata_dpt_depth_v2.ckpt")
        if args.backbone == "vitl16_384" and args.task == "depth":
            weights = "./pretrained_models/omnidata_dpt_depth_v1.ckpt"  # the DPT-Large depth model (demo.py:80-81)
    model = build_model(args.task, weights=weights, random_weights=args.random_weights, dtype=args.dtype, max_batch=1,
                        backbone=args.backbone)
    model.to(device)

    def save_outputs(img_path, output_file_name):
        with torch.no_grad():
            save_path = os.path.join(args.output_path, f"{output_file_name}_{args.task}.png")
            print(f"Reading input {img_path} ...")
            img = Image.open(img_path)
            # Resize/CenterCrop/ToTensor(/Normalize) run on the GPU from the raw uint8 pixels (bit-identical to
            # the PIL/torchvision path of the reference; RGBA and other modes fall back to PIL inside)
            img_tensor = pp.image_to_input_gpu(img, args.task, device)
            pp.rgb_preview(img).save(os.path.join(args.output_path, f"{output_file_name}_rgb.png"))
            output = model(img_tensor).clamp(min=0, max=1)
            if args.task == "depth":
                d512 = pp.depth_to_512_gpu(output)          # bicubic 384->512, clamp, 1-x on the GPU
                Image.fromarray(pp.colorize_viridis(d512.cpu().numpy())).save(save_path)
            else:
                Image.fromarray(pp.normal_to_u8_gpu(output[0]).cpu().numpy()).save(save_path)
            print(f"Writing output {save_path} ...")

    img_path = Path(args.img_path)
    if img_path.is_file():
        save_outputs(args.img_path, os.path.splitext(os.path.basename(args.img_path))[0])
    elif img_path.is_dir():
        for f in glob.glob(args.img_path + "/*"):
            save_outputs(f, os.path.splitext(os.path.basename(f))[0])
    else:
        print("invalid file path!")
        sys.exit()


if __name__ == "__main__":
    main()
