import sys, os, ctypes
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
if mode == "torch_only":
    import torch; print("torch only:", torch.cuda.is_available(), torch.cuda.device_count()); 
elif mode == "lib_first":
    L = ctypes.CDLL("go-pocket-tts_amd/libptts_hip.so")
    n = ctypes.c_int(0); hip = ctypes.CDLL("libamdhip64.so"); print("hipGetDeviceCount rc", hip.hipGetDeviceCount(ctypes.byref(n)), n.value)
    import torch; print("lib first -> torch:", torch.cuda.is_available())
elif mode == "torch_first":
    import torch; print("torch:", torch.cuda.is_available()); x = torch.zeros(4, device="cuda"); 
    L = ctypes.CDLL("go-pocket-tts_amd/libptts_hip.so")
    import ptts_amd; pkg = ptts_amd.load()
    import numpy as np
    print("op after torch:", pkg.runtime.op_linear(np.ones((1,4),np.float32), np.ones((2,4),np.float32)))
os.system("grep -E 'amdhip|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
