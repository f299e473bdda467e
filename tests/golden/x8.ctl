GENERAL-INFO-START

	seq-file            x8.seq
	trace-file          x8.trace
	locus-mut-rate          CONST
	num-loci            10
	random-seed         12345
	mcmc-iterations	  24
	iterations-per-log  8
	logs-per-line       10

	find-finetunes		FALSE
	finetune-coal-time	0.01		
	finetune-mig-time	0.3		
	finetune-theta		0.04
	finetune-mig-rate	0.02
	finetune-tau		0.0000008
	finetune-mixing		0.003

	tau-theta-print		10000.0
	tau-theta-alpha		1.0
	tau-theta-beta		10000.0

	mig-rate-print		0.001
	mig-rate-alpha		0.002
	mig-rate-beta		0.0000000400

GENERAL-INFO-END

CURRENT-POPS-START	

	POP-START
		name		A
		samples		s0 d
	POP-END

	POP-START
		name		B
		samples		s1 d
	POP-END

	POP-START
		name		C
		samples		s2 d
	POP-END

	POP-START
		name		D
		samples		s3 d
	POP-END

	POP-START
		name		E
		samples		s4 d
	POP-END

	POP-START
		name		F
		samples		s5 d
	POP-END

	POP-START
		name		G
		samples		s6 d
	POP-END

	POP-START
		name		H
		samples		s7 d
	POP-END

	POP-START
		name		I
		samples		s8 d
	POP-END

	POP-START
		name		J
		samples		s9 d
	POP-END

	POP-START
		name		K
		samples		s10 d
	POP-END

	POP-START
		name		L
		samples		s11 d
	POP-END

	POP-START
		name		M
		samples		s12 d
	POP-END

	POP-START
		name		N
		samples		s13 d
	POP-END

	POP-START
		name		O
		samples		s14 d
	POP-END

	POP-START
		name		P
		samples		s15 d
	POP-END

CURRENT-POPS-END

ANCESTRAL-POPS-START

	POP-START
		name			AB
		children		A		B
		tau-initial	0.000005000
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABC
		children		AB		C
		tau-initial	0.000006500
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCD
		children		ABC		D
		tau-initial	0.000008450
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDE
		children		ABCD		E
		tau-initial	0.000010985
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEF
		children		ABCDE		F
		tau-initial	0.000014281
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFG
		children		ABCDEF		G
		tau-initial	0.000018565
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGH
		children		ABCDEFG		H
		tau-initial	0.000024134
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHI
		children		ABCDEFGH		I
		tau-initial	0.000031374
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJ
		children		ABCDEFGHI		J
		tau-initial	0.000040787
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJK
		children		ABCDEFGHIJ		K
		tau-initial	0.000053022
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJKL
		children		ABCDEFGHIJK		L
		tau-initial	0.000068929
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJKLM
		children		ABCDEFGHIJKL		M
		tau-initial	0.000089608
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJKLMN
		children		ABCDEFGHIJKLM		N
		tau-initial	0.000116490
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDEFGHIJKLMNO
		children		ABCDEFGHIJKLMN		O
		tau-initial	0.000151438
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			root
		children		ABCDEFGHIJKLMNO		P
		tau-initial	0.000757188
		tau-beta		20000.0	
		finetune-tau			0.00000286
	POP-END

ANCESTRAL-POPS-END

MIG-BANDS-START	
	BAND-START		
       source  A
       target  B
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  B
       target  C
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  C
       target  D
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  D
       target  E
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  E
       target  F
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  F
       target  G
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  G
       target  H
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  H
       target  I
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  B
       target  A
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  C
       target  B
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  D
       target  C
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  E
       target  D
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  F
       target  E
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  G
       target  F
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  H
       target  G
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  I
       target  H
       mig-rate-print 0.1
	BAND-END

MIG-BANDS-END
