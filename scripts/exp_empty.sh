for sp in 4 8 32 128; do timeout -k 10 60 ./tools/decode_bench --bs 64 --ctx 2048 --ps 1 --hq 16 --hkv 8 --chunk 512 --splits $sp --layers 8 --steps 40 | cut -c1-160; done
for sp in 8 16 32; do timeout -k 10 60 ./tools/decode_bench --bs 256 --ctx 4096 --ps 16 --hq 32 --hkv 8 --chunk 512 --splits $sp --layers 4 --steps 20 | cut -c1-160; done
for sp in 4 8 32; do timeout -k 10 60 ./tools/decode_bench --bs 256 --ctx 4096 --ps 16 --hq 32 --hkv 8 --chunk 1024 --splits $sp --layers 4 --steps 20 | cut -c1-160; done
