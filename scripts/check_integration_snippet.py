import re, torch
src = open("INTEGRATION.md").read()
code = re.search(r"## 2\. Binding.*?```python\n(.*?)```", src, re.S).group(1)
exec(code)
x = torch.randn(2, 16, 16, 32, device="cuda")
w = torch.randn(64, 32, 3, 3, device="cuda") * 0.1
y = conv2d_forward_nhwc(x, w)
ref = torch.nn.functional.conv2d(x.cpu().permute(0, 3, 1, 2), w.cpu(), None, 1, 1).permute(0, 2, 3, 1)
print("max err", (y.cpu() - ref).abs().max().item())
