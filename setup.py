from setuptools import find_packages, setup

setup(name="kokoro-ruslan-amd", version="0.1.0",
      description="MI355X-native engine for the Kokoro (RUSLAN) acoustic-model train step",
      packages=find_packages(include=["kokoro", "kokoro.*", "kokoro_ruslan_amd", "kokoro_ruslan_amd.*"]),
      package_data={"kokoro_ruslan_amd": ["libkokoro_hip.so", "csrc/*"]},
      entry_points={"console_scripts": ["kokoro-train=kokoro.cli.training:main"]})
