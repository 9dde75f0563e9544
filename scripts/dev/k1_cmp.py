import torch,sys
a=torch.load(sys.argv[1]); b=torch.load(sys.argv[2])
for k in a:
    d=(a[k]-b[k]).abs()
    print(k, "max", float(d.max()), "mean", float(d.mean()), "ref mean abs", float(a[k].abs().mean()))
