"""`opt`: the reference's edit-the-file configuration object (config.py:19-52) with identical attribute
names and defaults, plus the knobs the MI355X path adds (defaults leave reference behaviour unchanged)."""
JOINT = {"nyu": 14, "icvl": 16, "msra": 21, "hands17": 21}
STEP = {"nyu": 30, "icvl": 10, "msra": 10, "hands17": 5}
EPOCH = {"nyu": 40, "icvl": 40, "msra": 25, "hands17": 10}


class Config(object):
    gpu_id = 0
    exp_id = "nyu_hourglass"
    log_id = "dense"
    data_dir = "./data"
    dataset = "nyu"
    output_dir = "./output/"
    load_model = "./results/hourglass_1.pth"
    jt_num = JOINT[dataset]
    cube = [300, 300, 300]
    augment_para = [10, 0.1, 180]
    net = "hourglass_1"          # or 'resnet_18'
    downsample = 2
    img_size = 128
    batch_size = 32
    num_workers = 8
    max_epoch = EPOCH[dataset]
    loss_type = "MyL1Loss"
    dense_weight = 1.0
    coord_weight = 0
    kernel_size = 0.4            # 0.4 for hourglass, 1 for resnet
    lr = 1e-3
    optimizer = "adam"
    scheduler = "step"
    step = STEP[dataset]
    weight_decay = 0
    print_freq = 100
    vis_freq = 1
    # ---- additions of the MI355X path ----
    use_hipgraph = True          # replay each step as one hipGraph
    world_size = 1               # data-parallel ranks (one process per GPU, RCCL)


opt = Config()
