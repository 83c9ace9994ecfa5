for o in "roi_align.fold_order=0,roi_align.units_per_wave=1" "roi_align.fold_order=0,roi_align.units_per_wave=2" "roi_align.fold_order=0,roi_align.units_per_wave=3" "roi_align.fold_order=1" "roi_align.fold_order=0,roi_align.units_per_wave=1" "roi_align.fold_order=0,roi_align.units_per_wave=2" "roi_align.fold_order=0,roi_align.units_per_wave=4"; do
TVMI_SET_OPTIONS=$o python tools/roi_knock.py "$o" 7 2>&1 | tail -1
done
