# builds a variant of the library for scripts/gpu_ab*.sh: bash scripts/build_variant.sh NAME "<extra hipcc flags>"
#   -> sphfluidsimulation_amd/lib/variants/NAME.so; the default library (and its objects' time stamps) are put back afterwards, so a
#   later `python -m sphfluidsimulation_amd.build` without --force cannot mistake a variant for the default build
set -e
L=sphfluidsimulation_amd/lib
mkdir -p $L/variants
[ -f $L/libsph_hip.so ] && cp -p $L/libsph_hip.so $L/.default_backup.so
SPH_HIPCC_EXTRA="$2" python -m sphfluidsimulation_amd.build --force > /dev/null
cp $L/libsph_hip.so $L/variants/$1.so
if [ -f $L/.default_backup.so ]; then mv $L/.default_backup.so $L/libsph_hip.so; else rm -f $L/libsph_hip.so; fi
rm -rf $L/obj
echo "built variants/$1.so [$2]"
